"""tools/e2e_quick.py [reads = 16000000]: `trew short 5 32` on a plain FASTQ of the bench workload in /dev/shm, text batches
(device pack) against --host_pack, 8 and 16 threads; prints the [trew] --stats lines."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from trew_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
d = tempfile.mkdtemp(prefix="trew_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "e2e.fastq")
bench.write_fastq(path, capi, n, 150)
trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
outs = {}
try:
    for t in (16, 16):
        for extra in ([], ["--host_pack"]):
            r = subprocess.run([trew, "short", "5", "32", path, "-t", str(t), "--stats", *extra], capture_output=True, text=True)
            outs[tuple(extra)] = r.stdout
            print("threads", t, " ".join(extra) or "device pack", "|", " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) or r.stderr[-300:])
    print("CSV identical:", outs[()] == outs[("--host_pack",)], len(outs[()]))
finally:
    os.remove(path)
    os.rmdir(d)
