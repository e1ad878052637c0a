#!/usr/bin/env python3
"""BGZF input end to end against its zlib ceiling (profiles/r04/README.md): writes N synthetic 150-bp reads as FASTQ, compresses them
into BGZF members (64 KiB blocks, 'BC' subfield, zlib level 1 -- the level only changes the file's size), then times
  (a) bgzf_cat FILE T > /dev/null         the parallel member inflate alone (BgzfReader, T inflate threads), nothing parsed
  (b) trew short 5 32 FILE -t 16 --stats  the whole path: inflate -> serial reader -> pack -> device scan -> CSV
Usage: tools/e2e_bgzf.py [reads = 8000000]"""
import multiprocessing as mp
import os
import re
import struct
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def member(data):
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    bsize = 18 + len(body) + 8 - 1
    head = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def main():
    import bench
    from trew_amd import capi

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    d = tempfile.mkdtemp(prefix="trew_bgzf_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        plain = os.path.join(d, "r.fastq")
        bench.write_fastq(plain, capi, n, 150)
        text = open(plain, "rb").read()
        os.unlink(plain)
        blocks = [text[i:i + 0xFF00] for i in range(0, len(text), 0xFF00)]
        with mp.Pool(16) as pool:
            members = pool.map(member, blocks, chunksize=256)
        gz = os.path.join(d, "r.fastq.gz")
        with open(gz, "wb") as f:
            for m in members:
                f.write(m)
            f.write(member(b""))
        bases = n * 150
        print("%d reads, %.2f GB of text, %.2f GB BGZF" % (n, len(text) / 1e9, os.path.getsize(gz) / 1e9))
        del text, blocks, members
        cat = os.path.join(ROOT, "trew_amd", "bin", "bgzf_cat")
        for t in (4, 8, 15, 16, 24):
            best = None
            for rep in range(2):
                t0 = time.perf_counter()
                with open(os.devnull, "wb") as nul:
                    subprocess.run([cat, gz, str(t), str(4 << 20)], stdout=nul, check=True)
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            print("inflate only, %2d threads: %.3f s = %.2f Gbases/s" % (t, best, bases / best / 1e9))
        trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
        for rep in range(2):
            r = subprocess.run([trew, "short", "5", "32", gz, "-t", "16", "--stats"], capture_output=True, text=True)
            m = re.search(r"([0-9.]+) s, ([0-9.]+) Gbases/s end-to-end", r.stderr)
            print("trew -t 16:", m.group(0) if m else r.stderr[-300:])
            if rep == 1:
                print(r.stderr[-1500:])
        r = subprocess.run([trew, "short", "5", "32", gz, "-t", "16", "--stats", "--serial_reader"], capture_output=True, text=True)
        m = re.search(r"([0-9.]+) s, ([0-9.]+) Gbases/s end-to-end", r.stderr)
        print("trew -t 16 --serial_reader:", m.group(0) if m else r.stderr[-300:])
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
