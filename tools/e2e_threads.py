"""tools/e2e_threads.py [reads = 48000000] T1 T2 ...: `trew short 5 32 -t T --stats` on a plain FASTQ of the bench workload in
/dev/shm for each thread count (twice, the second is reported as well): does the end-to-end rate still grow with threads?"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from trew_amd import capi  # noqa: E402

args = sys.argv[1:]
n = int(args.pop(0)) if args else 48_000_000
threads = [int(x) for x in args] or [8, 16]
d = tempfile.mkdtemp(prefix="trew_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "e2e.fastq")
bench.write_fastq(path, capi, n, 150)
trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
print("cpus allowed:", len(os.sched_getaffinity(0)), flush=True)
try:
    for rep in range(2):
        for t in threads:
            r = subprocess.run([trew, "short", "5", "32", path, "-t", str(t), "--stats"], capture_output=True, text=True)
            print("-t %-3d %s" % (t, " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) or r.stderr[-300:]), flush=True)
finally:
    os.remove(path)
    os.rmdir(d)
