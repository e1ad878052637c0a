"""tools/pair_counters.py: fall-back counters of the pair kernel's group pass on 2 M synthetic pairs (group_routed counts the pairs
whose rows all came back but whose chains needed a record at a k the row did not keep -- they take the intent list and flush())."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trew_amd as T  # noqa: E402

n = 2_000_000
t = T.TrewHip(mode=T.MODE_PAIR, n_slots=1, max_batch_words=16, max_batch_reads=2 * n, table_log2_slots=20)
d = t.malloc(2 * n * 60 + 64)
t.synth_pair_device(20250218, 0, n, 150, d)
b = t.device_uniform_batch(d, 2 * n, 150)
t.reset_tables()
t.submit(b, 0)
t.wait(0)
print("timing", t.last_timing(0), "counters", t.debug_counters())
