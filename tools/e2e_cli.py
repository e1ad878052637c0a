"""End-to-end CLI throughput on an uncompressed FASTQ in page cache (PCIe-inclusive; decode + pack bound)."""
import subprocess, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trew_amd import capi
n = 8_000_000
buf, st, nd = capi.synth_short_ascii(20250218, 0, n, 150)
path = "/tmp/e2e.fastq"
b = np.frombuffer(buf, dtype=np.uint8).reshape(n, 151)
rec = np.zeros((n, 3 + 151 + 2 + 151), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:154] = b
rec[:, 154:156] = np.frombuffer(b"+\n", dtype=np.uint8)
rec[:, 156:306] = ord("I")
rec[:, 306] = ord("\n")
rec.tofile(path)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for t in (2, 4, 8, 16):
    r = subprocess.run([os.path.join(root, "trew_amd/bin/trew"), "short", "5", "32", path, "-t", str(t), "--stats"], capture_output=True, text=True)
    print("threads", t, r.stderr.strip().splitlines()[-1] if r.stderr else r.returncode)
