"""tools/e2e_blocks.py [reads]: e2e throughput of `trew short 5 32 -t 16` against the scan block size (TREW_SCAN_BLOCK_KIB)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from trew_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
d = tempfile.mkdtemp(prefix="trew_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "e2e.fastq")
bench.write_fastq(path, capi, n, 150)
trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
try:
    for kib in (4096, 4096, 2048, 1024, 512, 256, 1024):
        env = dict(os.environ, TREW_SCAN_BLOCK_KIB=str(kib))
        r = subprocess.run([trew, "short", "5", "32", path, "-t", "16", "--stats"], capture_output=True, text=True, env=env)
        print("block KiB", kib, "|", " | ".join(x.split(": ", 1)[-1][:230] for x in r.stderr.strip().splitlines() if x.startswith("[trew]")))
finally:
    os.remove(path)
    os.rmdir(d)
