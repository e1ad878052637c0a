"""Interleaved A/B timing of bench variants in ONE process (cdna_hip_programming.md rule 24):
python tools/ab_flags.py 0 2 ...   (TREW_FLAG_* values; each gets its own context on the same reads)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trew_amd as T

flags = [int(x) for x in sys.argv[1:]] or [0, 2]
n, L = 10_000_000, 150
ctxs = [T.TrewHip(mode=T.MODE_SHORT, n_slots=1, max_batch_words=16, max_batch_reads=n, table_log2_slots=20, flags=f) for f in flags]
d = ctxs[0].malloc(n * 60 + 64)
ctxs[0].synth_short_device(20250218, 0, n, L, d)
b = ctxs[0].device_uniform_batch(d, n, L)
res = {f: [] for f in flags}
for rnd in range(6):
    for f, t in zip(flags, ctxs):
        for _ in range(4):
            t.submit(b, 0)
        t.wait(0)
        a, e, _ = t.last_timing(0, want_flagged=False)
        if rnd:
            res[f].append((a, e))
for f in flags:
    a = np.array(res[f])
    print("flags=%d filter median %.4f min %.4f | exact median %.4f min %.4f ms" % (f, np.median(a[:, 0]), a[:, 0].min(), np.median(a[:, 1]), a[:, 1].min()))
