"""tools/flag_diff.py PASSES [short|pair|long]: the same device batch (12 M reads, 6 M pairs or 1 M long reads) PASSES times on
alternating slots; the worklist of every pass is compared with the first one as a multiset -- prints units that are missing,
extra or duplicated."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import trew_amd as T  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else "short"
n, L, seed = 12_000_000, 150, 20250218
stride = 3 * ((L + 31) // 32)
dev_mode = {"short": T.MODE_SHORT, "pair": T.MODE_PAIR, "long": T.MODE_LONG}[mode]
with T.TrewHip(mode=dev_mode, n_slots=2, max_batch_reads=n, max_batch_words=16, table_log2_slots=20) as t:
    if mode == "long":
        batch, to_free, _ = t.synth_long_device(seed, 0, 1_000_000)
    else:
        d = t.malloc(n * stride * 4 + 64)
        if mode == "pair":
            t.synth_pair_device(seed, 5_000_000_000, n // 2, L, d)
        else:
            t.synth_short_device(seed, 5_000_000_000, n, L, d)
        batch = t.device_uniform_batch(d, n, L)
    ref = None
    bad = 0
    for rep in range(passes):
        t.reset_tables()
        t.submit(batch, rep & 1)
        t.wait(rep & 1)
        if os.environ.get("FLAG_DIFF_TABLES"):  # the tables as well (slower)
            tab = t.collect()
            if rep == 0:
                ref_tab = tab
            elif tab != ref_tab:
                bad += 1
                print("pass", rep, "tables differ")
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
