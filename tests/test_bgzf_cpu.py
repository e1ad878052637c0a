"""BGZF (block gzip) reader of the host: parallel inflate must return exactly the bytes gzip returns."""
import gzip
import os
import random
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGZF_CAT = os.path.join(ROOT, "trew_amd", "bin", "bgzf_cat")


def write_bgzf(path, data, block=0xFF00, level=6):
    """Minimal BGZF writer (SAM spec section 4.1): independent gzip members with a 'BC' extra subfield."""
    with open(path, "wb") as f:
        chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]  # the empty member is the EOF marker
        for c in chunks:
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = co.compress(c) + co.flush()
            bsize = 18 + len(body) + 8 - 1
            f.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize))
            f.write(body)
            f.write(struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c) & 0xFFFFFFFF))


def fastq(n, seed):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        L = rnd.choice([36, 100, 150, 151, 250])
        s = "".join(rnd.choice("ACGTN" if rnd.random() < 0.05 else "ACGT") for _ in range(L))
        out.append("@r%d\n%s\n+\n%s\n" % (i, s, "I" * L))
    return "".join(out).encode()


@pytest.mark.skipif(not os.path.exists(BGZF_CAT), reason="host tools not built")
@pytest.mark.parametrize("block,threads,read_size", [(0xFF00, 4, 1 << 20), (1000, 3, 4096), (0xFF00, 1, 77), (50, 8, 1 << 22)])
def test_bgzf_reader_matches_gzip(tmp_path, block, threads, read_size):
    data = fastq(3000, block + threads)
    p = str(tmp_path / "reads.fastq.gz")
    write_bgzf(p, data, block=block)
    assert gzip.open(p, "rb").read() == data  # a BGZF file is a valid multi-member gzip file
    got = subprocess.run([BGZF_CAT, p, str(threads), str(read_size)], capture_output=True, timeout=120)
    assert got.returncode == 0, got.stderr
    assert got.stdout == data


@pytest.mark.skipif(not os.path.exists(BGZF_CAT), reason="host tools not built")
def test_bgzf_reader_rejects_damage(tmp_path):
    data = fastq(500, 9)
    p = str(tmp_path / "reads.fastq.gz")
    write_bgzf(p, data, block=4000)
    raw = bytearray(open(p, "rb").read())
    # plain gzip is not BGZF: the CLI keeps using gzread for it
    q = str(tmp_path / "plain.gz")
    with gzip.open(q, "wb") as f:
        f.write(data)
    assert subprocess.run([BGZF_CAT, q], capture_output=True).returncode == 3
    # flip a byte inside the deflate stream of the third member: CRC / inflate error, not silent garbage
    off = 0
    for _ in range(2):
        off += struct.unpack("<H", raw[off + 16:off + 18])[0] + 1
    raw[off + 40] ^= 0x5A
    bad = str(tmp_path / "bad.gz")
    open(bad, "wb").write(bytes(raw))
    r = subprocess.run([BGZF_CAT, bad, "4"], capture_output=True, timeout=60)
    assert r.returncode == 1 and b"error" in r.stderr
    # truncated file
    trunc = str(tmp_path / "trunc.gz")
    open(trunc, "wb").write(bytes(raw[: len(raw) // 2]))
    open(trunc, "r+b").close()
    r = subprocess.run([BGZF_CAT, p, "4"], capture_output=True, timeout=60)
    assert r.returncode == 0
    r = subprocess.run([BGZF_CAT, trunc, "4"], capture_output=True, timeout=60)
    assert r.returncode == 1


@pytest.mark.skipif(not os.path.exists(BGZF_CAT), reason="host tools not built")
def test_bgzf_then_plain_gzip_and_foreign_subfields(tmp_path):
    """What zlib's gzread accepts, the parallel reader accepts: `cat a.bgz b.gz` (BGZF members followed by an ordinary
    gzip member -- the tail goes through zlib), and BGZF members whose BC subfield is not the first extra subfield."""
    a, b = fastq(800, 21), fastq(700, 22)
    pa, pb, mixed = str(tmp_path / "a.gz"), str(tmp_path / "b.gz"), str(tmp_path / "mixed.gz")
    write_bgzf(pa, a, block=5000)
    with gzip.open(pb, "wb") as f:
        f.write(b)
    open(mixed, "wb").write(open(pa, "rb").read() + open(pb, "rb").read())
    assert gzip.open(mixed, "rb").read() == a + b
    for threads, read_size in ((4, 1 << 20), (2, 999)):
        got = subprocess.run([BGZF_CAT, mixed, str(threads), str(read_size)], capture_output=True, timeout=120)
        assert got.returncode == 0, got.stderr
        assert got.stdout == a + b
    # BC after another subfield (RFC 1952 allows any order)
    data = fastq(300, 23)
    p = str(tmp_path / "sub.gz")
    with open(p, "wb") as f:
        for c in [data[i:i + 3000] for i in range(0, len(data), 3000)] + [b""]:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = co.compress(c) + co.flush()
            extra_other = b"XY" + struct.pack("<H", 3) + b"abc"
            xlen = len(extra_other) + 6
            bsize = 12 + xlen + len(body) + 8 - 1
            f.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", xlen) + extra_other + b"BC" + struct.pack("<HH", 2, bsize))
            f.write(body)
            f.write(struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c) & 0xFFFFFFFF))
    assert gzip.open(p, "rb").read() == data
    got = subprocess.run([BGZF_CAT, p, "3"], capture_output=True, timeout=120)
    assert got.returncode == 0 and got.stdout == data
    # a trailer that claims more than 64 KiB is refused before anything is allocated
    raw = bytearray(open(pa, "rb").read())
    first = struct.unpack("<H", raw[16:18])[0] + 1
    raw[first - 4:first] = struct.pack("<I", 0x7FFFFFFF)
    bad = str(tmp_path / "isize.gz")
    open(bad, "wb").write(bytes(raw))
    r = subprocess.run([BGZF_CAT, bad, "2"], capture_output=True, timeout=60)
    assert r.returncode == 1 and b"64 KiB" in r.stderr
