"""tools/e2e_numa.py [reads]: where do the worker threads of `trew` run?  The same end-to-end run under different taskset masks
(whole machine, each NUMA node, one CPU per L3) -- prints lscpu's NUMA lines, the GPU's node and the --stats lines."""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from trew_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48_000_000
print(subprocess.run("lscpu | grep -i 'numa\\|socket\\|L3'", shell=True, capture_output=True, text=True).stdout)
for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(f, open(f).read().strip())
print("self allowed:", sorted(os.sched_getaffinity(0))[:4], "...", len(os.sched_getaffinity(0)))
d = tempfile.mkdtemp(prefix="trew_e2e_", dir="/dev/shm")
path = os.path.join(d, "e2e.fastq")
bench.write_fastq(path, capi, n, 150)
trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
masks = ["", "0-63,128-191", "64-127,192-255", "0-127", "0-127:8", "0-31,128-159", "32-63,160-191"]
try:
    for rep in range(2):
        for m in masks:
            cmd = (["taskset", "-c", m] if m else []) + [trew, "short", "5", "32", path, "-t", "16", "--stats"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            line = [x for x in r.stderr.strip().splitlines() if "Gbases/s" in x]
            print("%-18s %s" % (m or "all", line[0].split(", ")[3] if line else r.stderr[-200:]), flush=True)
finally:
    os.remove(path)
    os.rmdir(d)
