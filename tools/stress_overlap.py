"""tools/stress_overlap.py N PARTS PASSES [mode = short|pair]: the same N units as PARTS device batches queued round-robin on TWO
slots without waiting in between (so an exact kernel usually finds the other slot busy and takes half of its wave slots,
trew_kernels.hip::launch_exact), PASSES times; every pass must give the tables of a first pass that ran the parts one at a
time on one slot.  Prints the passes that differ."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trew_amd as T  # noqa: E402

n, parts, passes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "short"
L, seed = 150, 20250218
stride = 3 * ((L + 31) // 32)
pair = mode == "pair"
rp = 2 if pair else 1  # reads per unit


def digest(tabs):
    h = hashlib.sha256()
    for name in sorted(tabs):
        for key in sorted(tabs[name]):
            h.update(("%s %d %d %d;" % (name, key[0], key[1], tabs[name][key])).encode())
    return h.hexdigest()[:16], sum(sum(v.values()) for v in tabs.values())


per = n // parts
with T.TrewHip(mode=T.MODE_PAIR if pair else T.MODE_SHORT, n_slots=2, max_batch_reads=rp * per, max_batch_words=16, table_log2_slots=22) as t:
    bufs = []
    for p in range(parts):
        d = t.malloc(rp * per * stride * 4 + 64)
        if pair:
            t.synth_pair_device(seed, p * per, per, L, d)
        else:
            t.synth_short_device(seed, p * per, per, L, d)
        bufs.append(d)
    t.reset_tables()
    for d in bufs:  # reference pass: one at a time, one slot
        t.submit(t.device_uniform_batch(d, rp * per, L), 0)
        t.wait(0)
    ref = digest(t.collect())
    print("serial reference", ref, flush=True)
    bad = 0
    for rep in range(passes):
        t.reset_tables()
        for i, d in enumerate(bufs):
            t.submit(t.device_uniform_batch(d, rp * per, L), (i + rep) & 1)
        t.wait(0)
        t.wait(1)
        got = digest(t.collect())
        if got != ref:
            bad += 1
            print("pass", rep, "differs:", got, flush=True)
    print("passes", passes, "different", bad)
    sys.exit(1 if bad else 0)
