"""Does running the exact kernel of pass i beside the filter of pass i+1 (two slots = two HIP streams) raise throughput?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trew_amd as T
n, L = 10_000_000, 150
t = T.TrewHip(mode=T.MODE_SHORT, n_slots=2, max_batch_words=16, max_batch_reads=n, table_log2_slots=20)
d = t.malloc(n * 60 + 64)
t.synth_short_device(20250218, 0, n, L, d)
b = t.device_uniform_batch(d, n, L)
for s in (0, 1):
    t.submit(b, s); t.wait(s)
t.collect_rows()
for label, slots in (("one stream", [0]), ("two streams", [0, 1]), ("one stream", [0]), ("two streams", [0, 1])):
    t0 = time.perf_counter()
    K = 40
    for i in range(K):
        t.submit(b, slots[i % len(slots)])
    for s in slots: t.wait(s)
    dt = time.perf_counter() - t0
    print(label, "%.3f ms/pass  %.1f Gbases/s" % (dt / K * 1e3, n * L * K / dt / 1e9), [t.last_timing(s, want_flagged=False)[:2] for s in slots])
