"""tests/harness/filter_selectivity.py [reads = 300000]: how selective is the prefilter on the bench workload?  Runs the filter
alone (trew_hip_filter_masks) on synthetic 150-bp reads and checks every candidate (read, half, k) it keeps against the
oracle's exact MAX / COUNT: how many flagged reads have no passing k at all (false positives that the exact kernel pays
for), split by whether the read carries an N, and how far their best ratio is from the threshold.  A diagnostic, not a test."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
import trew_amd as T  # noqa: E402
from trew_amd import capi  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
n = 150
buf, st, nd = capi.synth_short_ascii(20250218, 0, n_reads, n)
reads = [buf[s:e + 1] for s, e in zip(st, nd)]
words, offs, lens = capi.pack_reads(reads)
stride = 3 * ((n + 31) // 32)
with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=n_reads + 8, max_batch_words=len(words) + 64) as t:
    b = capi.Batch(words.ctypes.data, len(words), None, None, n, stride, n_reads, 0, 0)
    cand = t.filter_masks(b, 3)
p = O.OracleParams()
segs = [(0, n // 2), (n - (n + 1) // 2, n)]
flagged = np.nonzero((cand[:, 0] | cand[:, 1]) != 0)[0]
print("reads", n_reads, "flagged", len(flagged), "= %.3f %%" % (100.0 * len(flagged) / n_reads))
kinds = collections.Counter()
best_fp = []
cand_k = collections.Counter()
for i in flagged:
    r = reads[i]
    has_n = any(c not in b"ACGTacgt" for c in r)
    true_any = False
    best = 0.0
    for slot, (a, e) in enumerate(segs):
        m = int(cand[i, slot])
        if not m:
            continue
        stt = O.segment_stats(p, r[a:e], 5, 32)
        for k, (cnt, mx, _) in stt.items():
            if (m >> (k - 1)) & 1:
                ratio = mx / cnt if cnt else 0.0
                best = max(best, ratio)
                ok = cnt and ratio >= 0.5
                cand_k[(k, bool(ok))] += 1
                true_any = true_any or bool(ok)
    kinds[("N" if has_n else "clean", "true" if true_any else "false")] += 1
    if not true_any:
        best_fp.append((best, has_n))
for key in sorted(kinds):
    print(key, kinds[key])
bf = np.array([x[0] for x in best_fp])
if len(bf):
    print("false positives: best exact ratio quantiles (threshold 0.5):", np.quantile(bf, [0.1, 0.25, 0.5, 0.75, 0.9, 0.99]).round(3))
    bn = np.array([x[0] for x in best_fp if x[1]])
    bc = np.array([x[0] for x in best_fp if not x[1]])
    if len(bn):
        print("  with N:", len(bn), np.quantile(bn, [0.1, 0.5, 0.9]).round(3))
    if len(bc):
        print("  clean :", len(bc), np.quantile(bc, [0.1, 0.5, 0.9]).round(3))
print("candidate (k, passes) counts:", sorted(cand_k.items()))
