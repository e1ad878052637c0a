"""End-to-end CLI throughput on an uncompressed FASTQ in page cache (PCIe-inclusive; decode + pack bound)."""
import subprocess, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trew_amd import capi
n = 32_000_000  # 9.8 GB of FASTQ text: at 15 Gbases/s the 8 M-read file of round 1 is over in 80 ms
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
    print("threads", t, " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) if r.stderr else r.returncode)

# paired files (config 3's shape): two plain FASTQ files, mates matched by read index
npair = 16_000_000
b1, b2, _, _ = capi.synth_pair_ascii(20250218, 0, npair, 150)
for name, bb in (("/tmp/e2e_r1.fastq", b1), ("/tmp/e2e_r2.fastq", b2)):
    bm = np.frombuffer(bb, dtype=np.uint8).reshape(npair, 151)
    rp = np.zeros((npair, 3 + 151 + 2 + 151), dtype=np.uint8)
    rp[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rp[:, 3:154] = bm
    rp[:, 154:156] = np.frombuffer(b"+\n", dtype=np.uint8)
    rp[:, 156:306] = ord("I")
    rp[:, 306] = ord("\n")
    rp.tofile(name)
    del rp, bm
del b1, b2
for t in (8, 16):
    for extra in ([], ["--serial_reader"]):
        r = subprocess.run([os.path.join(root, "trew_amd/bin/trew"), "short", "5", "32", "--paired_end", "--fq1", "/tmp/e2e_r1.fastq", "--fq2", "/tmp/e2e_r2.fastq",
                            "-t", str(t), "--stats", *extra], capture_output=True, text=True)
        print("pairs threads", t, " ".join(extra), " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) if r.stderr else r.returncode)

# long reads (config 4's shape): 150 k ONT-like reads, ~2.3 Gbases, 4.7 GB of FASTQ text
nlong = 150_000
lb, lst, lnd = capi.synth_long_ascii(20250218, 0, nlong)
with open("/tmp/e2e_long.fastq", "wb") as f:
    mv = memoryview(lb)
    for lo_i in range(0, nlong, 10000):  # large writes: the page cache then holds the file in large folios, like the other files
        parts = []
        for i in range(lo_i, min(nlong, lo_i + 10000)):
            seq = bytes(mv[int(lst[i]): int(lnd[i]) + 1])
            parts += [b"@r\n", seq, b"\n+\n", b"I" * len(seq), b"\n"]
        f.write(b"".join(parts))
del lb
for t in (8, 16):
    r = subprocess.run([os.path.join(root, "trew_amd/bin/trew"), "long", "5", "32", "/tmp/e2e_long.fastq", "-t", str(t), "--stats"], capture_output=True, text=True)
    print("long threads", t, " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) if r.stderr else r.returncode)

# the same file as plain gzip (one member: gzread on one thread, as the reference does) and as BGZF
# (independent 64 KiB members: inflated on several threads by host/bgzf_reader.hpp)
import gzip, zlib, struct
sub = rec[: 2_000_000].tobytes()  # 2 M reads: zlib at level 1 is slow enough already
gz = "/tmp/e2e_plain.fastq.gz"
with gzip.open(gz, "wb", compresslevel=1) as f:
    f.write(sub)
bg = "/tmp/e2e_bgzf.fastq.gz"
with open(bg, "wb") as f:
    for i in list(range(0, len(sub), 0xFF00)) + [None]:
        c = b"" if i is None else sub[i:i + 0xFF00]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(c) + co.flush()
        f.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1))
        f.write(body)
        f.write(struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c) & 0xFFFFFFFF))
for name, p, t in (("plain gzip", gz, 8), ("BGZF -t 8", bg, 8), ("BGZF -t 16", bg, 16)):
    r = subprocess.run([os.path.join(root, "trew_amd/bin/trew"), "short", "5", "32", p, "-t", str(t), "--stats"], capture_output=True, text=True)
    print(name, " | ".join(x for x in r.stderr.strip().splitlines() if x.startswith("[trew]")) if r.stderr else r.returncode)
