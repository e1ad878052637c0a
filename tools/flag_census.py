"""Who survives the prefilter?  2 M reads of the bench workload: candidate masks per read (trew_hip_filter_masks), classified by
what the generator made of the read (telomeric / junction / other, with or without N)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trew_amd as T
from trew_amd import capi

n, L = 2_000_000, 150
buf, st, nd = capi.synth_short_ascii(20250218, 0, n, L)
arr = np.frombuffer(buf, dtype=np.uint8).reshape(n, L + 1)[:, :L]
with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=n, max_batch_words=16) as t:
    d = t.malloc(n * 60 + 64)
    t.synth_short_device(20250218, 0, n, L, d)
    cand = t.filter_masks(t.device_uniform_batch(d, n, L), 3)
    t.free(d)
flag = (cand != 0).any(axis=1)
has_n = (arr == ord("N")).any(axis=1)
# motif content: matches of base i with base i+6 (period-6 autocorrelation)
per6 = (arr[:, :-6] == arr[:, 6:]).sum(axis=1)
kind = np.where(per6 >= 130, "telomeric", np.where(per6 >= 80, "junction", "other"))
print("flagged", int(flag.sum()), "of", n)
for k in ("telomeric", "junction", "other"):
    for hn in (False, True):
        m = (kind == k) & (has_n == hn)
        print("%-10s N=%d  reads %8d  flagged %7d" % (k, hn, int(m.sum()), int((flag & m).sum())))
oth = np.flatnonzero(flag & (kind == "other"))
print("per-slot flagged among 'other':", [(int((cand[oth, s] != 0).sum())) for s in range(3)])
for i in oth[:8]:
    ks = [[k + 1 for k in range(64) if (int(cand[i, s]) >> k) & 1] for s in range(3)]
    print(i, bytes(arr[i]).decode(), ks, "per6", int(per6[i]))
