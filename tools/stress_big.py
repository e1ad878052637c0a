"""tools/stress_big.py N FIRST PASSES: the same N-read device batch PASSES times (tables reset in between, slots alternating);
prints every pass whose table digest or flagged count differs from the first pass."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trew_amd as T  # noqa: E402

n, first, passes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L, seed = 150, 20250218
stride = 3 * ((L + 31) // 32)


def digest(tabs):
    h = hashlib.sha256()
    for name in sorted(tabs):
        for key in sorted(tabs[name]):
            h.update(("%s %d %d %d;" % (name, key[0], key[1], tabs[name][key])).encode())
    return h.hexdigest()[:16], sum(sum(v.values()) for v in tabs.values())


with T.TrewHip(mode=T.MODE_SHORT, n_slots=2, max_batch_reads=n, max_batch_words=16, table_log2_slots=22) as t:
    d = t.malloc(n * stride * 4 + 64)
    t.synth_short_device(seed, first, n, L, d)
    ref = None
    bad = 0
    for rep in range(passes):
        t.reset_tables()
        t.submit(t.device_uniform_batch(d, n, L), rep & 1)
        t.wait(rep & 1)
        got = (digest(t.collect()), int(t.last_timing(rep & 1)[2]))
        if ref is None:
            ref = got
            print("pass 0", got)
        elif got != ref:
            bad += 1
            print("pass", rep, "DIFFERS", got)
    print("passes", passes, "differing", bad, "lib", os.environ.get("TREW_HIP_LIB", "main"))
