"""tools/e2e_pair.py [pairs = 16000000] [extra trew args...]: `trew short 5 32 --paired_end -t 16 --stats` on two plain FASTQ files of
the bench's pair workload in /dev/shm; prints the [trew] --stats lines (twice: the first run warms the page tables)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from trew_amd import capi  # noqa: E402

args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 16_000_000
L = 150
d = tempfile.mkdtemp(prefix="trew_e2e_", dir="/dev/shm")
p1, p2 = os.path.join(d, "r1.fastq"), os.path.join(d, "r2.fastq")
with open(p1, "wb") as f1, open(p2, "wb") as f2:
    for lo in range(0, n, 1_000_000):
        m = min(1_000_000, n - lo)
        b1, b2, _, _ = capi.synth_pair_ascii(bench.SEED, lo, m, L)
        for buf, f in ((b1, f1), (b2, f2)):
            b = np.frombuffer(buf, dtype=np.uint8).reshape(m, L + 1)
            rec = np.zeros((m, 3 + (L + 1) + 2 + (L + 1)), dtype=np.uint8)
            rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
            rec[:, 3:3 + L + 1] = b
            rec[:, 4 + L:6 + L] = np.frombuffer(b"+\n", dtype=np.uint8)
            rec[:, 6 + L:6 + 2 * L] = ord("I")
            rec[:, 6 + 2 * L] = ord("\n")
            rec.tofile(f)
trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
first = None
try:
    for rep in range(2):
        for extra in ([], *([args] if args else [])):
            r = subprocess.run([trew, "short", "5", "32", "--paired_end", "--fq1", p1, "--fq2", p2, "-t", "16", "--stats", *extra], capture_output=True, text=True)
            first = r.stdout if first is None else first
            print(" ".join(extra) or "default", "same" if r.stdout == first else "DIFFERENT", "|", " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) or r.stderr[-300:], flush=True)
finally:
    os.remove(p1)
    os.remove(p2)
    os.rmdir(d)
