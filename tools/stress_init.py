"""tools/stress_init.py [N = 2000] [mode = pair|short]: N fresh contexts, each given one small batch at once and collected -- every one must
return the same tables (a context whose first batch raced with the table memsets of its own creation returned empty ones)."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trew_amd as T
from trew_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mode = sys.argv[2] if len(sys.argv) > 2 else "pair"
buf, st, nd = capi.synth_short_ascii(20250218, 0, 3000, 150)
reads = [buf[s:e + 1] for s, e in zip(st, nd)]
ref, bad = None, 0
for i in range(n):
    with T.TrewHip(mode=T.MODE_PAIR if mode == "pair" else T.MODE_SHORT, max_mer=64 if i & 1 else 32, max_batch_reads=len(reads) + 8, max_batch_words=1 << 20) as t:
        t.submit_reads(reads)
        t.wait()
        got = t.collect()
        t.reset_tables()
        t.submit_reads(reads)
        t.wait()
        got2 = t.collect()
    key = i & 1
    if ref is None:
        ref = {}
    if key not in ref:
        ref[key] = got
        assert sum(len(v) for v in got.values()) > 10
    if got != ref[key] or got2 != ref[key]:
        bad += 1
        print("context", i, "differs: rows", sum(len(v) for v in got.values()), sum(len(v) for v in got2.values()), flush=True)
print("contexts", n, "different", bad)
sys.exit(1 if bad else 0)
