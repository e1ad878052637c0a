import sys; sys.path.insert(0, "/root/repo")
import trew_amd as T
from trew_amd import capi
n=1_000_000
t = T.TrewHip(mode=T.MODE_SHORT, n_slots=1, max_batch_words=16, max_batch_reads=n, table_log2_slots=20)
d = t.malloc(n*60+64); t.synth_short_device(20250218, 0, n, 150, d); b = t.device_uniform_batch(d, n, 150)
t.submit(b,0); t.wait(0)
print(t.last_timing(0), t.debug_counters())
