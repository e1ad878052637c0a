"""tools/e2e_long.py [reads = 300000] [extra trew args...]: `trew long 5 32 -t 16 --stats` on a plain FASTQ file of the bench's
ONT-like long reads in /dev/shm; prints the [trew] --stats lines (twice)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from trew_amd import capi  # noqa: E402

args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 300_000
d = tempfile.mkdtemp(prefix="trew_e2e_", dir="/dev/shm")
path = os.path.join(d, "long.fastq")
total = 0
with open(path, "wb") as f:
    for lo in range(0, n, 20_000):
        m = min(20_000, n - lo)
        buf, st, nd = capi.synth_long_ascii(bench.SEED, lo, m)
        parts = []
        for s, e in zip(st, nd):
            seq = bytes(buf[s:e + 1])
            total += len(seq)
            parts.append(b"@r\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
        f.write(b"".join(parts))
print("reads", n, "bases", total, "file bytes", os.path.getsize(path), flush=True)
trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
first = None
try:
    for rep in range(2):
        for extra in ([], *([args] if args else [])):
            r = subprocess.run([trew, "long", "5", "32", path, "-t", "16", "--stats", *extra], capture_output=True, text=True)
            first = r.stdout if first is None else first
            print(" ".join(extra) or "default", "same" if r.stdout == first else "DIFFERENT", "|", " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) or r.stderr[-300:], flush=True)
finally:
    os.remove(path)
    os.rmdir(d)
