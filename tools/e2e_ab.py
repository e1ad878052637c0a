"""tools/e2e_ab.py [reads = 48000000] [ENV=VAL,ENV=VAL ...]...: `trew short 5 32 -t 16 --stats` on a plain FASTQ of the bench
workload in /dev/shm, once per environment setting given (the first run is always the default environment); each setting
runs twice.  Prints the [trew] --stats lines and whether every CSV equals the first."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from trew_amd import capi  # noqa: E402

args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 48_000_000
settings = [""] + args
d = tempfile.mkdtemp(prefix="trew_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "e2e.fastq")
bench.write_fastq(path, capi, n, 150)
trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
first = None
try:
    for rep in range(2):
        for st in settings:
            env = dict(os.environ)
            for kv in filter(None, st.split(",")):
                k, v = kv.split("=", 1)
                env[k] = v
            r = subprocess.run([trew, "short", "5", "32", path, "-t", "16", "--stats"], capture_output=True, text=True, env=env)
            first = r.stdout if first is None else first
            print("%-40s %s | %s" % (st or "default", "same" if r.stdout == first else "DIFFERENT", " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) or r.stderr[-300:]), flush=True)
finally:
    os.remove(path)
    os.rmdir(d)
