import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trew_amd as T
n = 1_000_000
t = T.TrewHip(mode=T.MODE_LONG, n_slots=1, max_batch_words=16, max_batch_reads=n, table_log2_slots=20)
b, to_free, _ = t.synth_long_device(20250218, 0, n)
for i in range(3):
    t.reset_tables(); t.submit(b, 0); t.wait(0)
    print(t.last_timing(0), t.debug_counters())
