"""tools/flag_diff.py PASSES: the same 12 M-read device batch PASSES times on alternating slots; the worklist of every pass is
compared with the first one as a multiset -- prints units that are missing, extra or duplicated."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import trew_amd as T  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n, L, seed = 12_000_000, 150, 20250218
stride = 3 * ((L + 31) // 32)
with T.TrewHip(mode=T.MODE_SHORT, n_slots=2, max_batch_reads=n, max_batch_words=16, table_log2_slots=20) as t:
    d = t.malloc(n * stride * 4 + 64)
    t.synth_short_device(seed, 5_000_000_000, n, L, d)
    ref = None
    bad = 0
    for rep in range(passes):
        t.reset_tables()
        t.submit(t.device_uniform_batch(d, n, L), rep & 1)
        t.wait(rep & 1)
        if os.environ.get("FLAG_DIFF_COLLECT"):
            t.collect()
            t.last_timing(rep & 1)
        wl = np.sort(np.asarray(t.debug_worklist(rep & 1), dtype=np.int64))
        if ref is None:
            ref = wl
            print("pass 0:", len(wl), "flagged, distinct", len(np.unique(wl)))
            continue
        if len(wl) != len(ref) or not np.array_equal(wl, ref):
            bad += 1
            c_ref, c_got = collections.Counter(ref.tolist()), collections.Counter(wl.tolist())
            extra = sorted((c_got - c_ref).elements())
            missing = sorted((c_ref - c_got).elements())
            dup = [u for u, k in c_got.items() if k > 1]
            print("pass", rep, "slot", rep & 1, "flagged", len(wl), "extra", extra[:12], "missing", missing[:12], "duplicated", dup[:12])
    print("passes", passes, "differing", bad)
