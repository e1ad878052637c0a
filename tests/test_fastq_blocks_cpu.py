"""host/fastq_blocks.hpp (the block-parallel FASTQ reader of the `trew` host) against the reference reader's rule --
a sequence line is the line whose closing newline makes `num & 3 == 2` (read_fastq_thread, kmer.cpp:987-1038) --
for block sizes down to one byte, several threads, CRLF, empty lines, lines longer than many blocks, and files
that do not end in a newline.  CPU only: the product's header is compiled with a small harness."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("harness") / "blocks_harness")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "harness", "blocks_harness.cpp")], check=True)
    return exe


def reference_rule(data: bytes):
    """(start, length) of every sequence line as the reference's newline counter sees them."""
    out, num, idx = [], 0, -1
    for i, ch in enumerate(data):
        if ch == 0x0A:
            num += 1
            if (num & 3) == 2:
                out.append((idx + 1, i - idx - 1))
            idx = i
    return out


def run(exe, path, block, threads, isa=None):
    env = dict(os.environ, TREW_SCAN_ISA=isa) if isa else None
    r = subprocess.run([exe, path, str(block), str(threads)], capture_output=True, text=True, check=True, env=env)
    return [tuple(int(x) for x in line.split()) for line in r.stdout.splitlines()]


def make_cases():
    rnd = random.Random(7)
    cases = {}
    recs = []
    for i in range(300):
        n = rnd.choice([0, 1, 5, 36, 150, 151, 400])
        seq = "".join(rnd.choice("ACGTN") for _ in range(n))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, seq, "I" * n))
    cases["regular"] = "".join(recs).encode()
    cases["crlf"] = "".join(recs).replace("\n", "\r\n").encode()
    cases["no_trailing_newline"] = "".join(recs).encode()[:-1]
    cases["ends_inside_sequence_line"] = ("".join(recs) + "@last\nACGTACGT").encode()
    cases["only_newlines"] = b"\n" * 1003
    cases["no_newline_at_all"] = b"ACGT" * 100
    cases["quality_starting_with_at"] = b"@a\nACGT\n+\n@@@@\n@b\nTTTT\n+\n@III\n" * 50
    long_line = "A" * 5000
    cases["long_lines"] = ("@x\n" + long_line + "\n+\n" + "I" * 5000 + "\n@y\nACGT\n+\nIIII\n").encode() * 3
    cases["one_byte"] = b"\n"
    return cases


@pytest.mark.parametrize("name", sorted(make_cases()))
def test_blocks_find_the_reference_lines(harness, tmp_path, name):
    data = make_cases()[name]
    path = str(tmp_path / "x.fastq")
    open(path, "wb").write(data)
    want = reference_rule(data)
    for block, threads in [(1, 3), (2, 2), (7, 4), (64, 8), (100, 1), (4096, 5), (1 << 22, 3)]:
        if block < 7 and len(data) > 20000:
            continue  # a block per byte is pointless on the big cases
        assert run(harness, path, block, threads) == want, (name, block, threads)
    # the newline scan has an AVX2 form (the default, above), an opt-in AVX-512 form and a memchr form
    for isa in ("avx512", "scalar"):
        assert run(harness, path, 4096, 2, isa) == want, (name, isa)


def test_paired_reader_helpers(harness, tmp_path):
    """LineIndex + LineCursor (the paired block reader): mates are located by read index -- lines 4r .. 4r+3 of each
    file -- from per-block newline counts, for any block size and any split of the read range into items; the two
    files have different line lengths, so their blocks do not line up."""
    rnd = random.Random(11)
    recs1, recs2 = [], []
    for i in range(700):
        n1, n2 = rnd.choice([0, 36, 150, 151]), rnd.choice([1, 75, 150, 250])
        recs1.append("@r%d/1\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGTN") for _ in range(n1)), "I" * n1))
        recs2.append("@read_number_%d/2 extra\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGT") for _ in range(n2)), "#" * n2))
    d1, d2 = "".join(recs1).encode(), "".join(recs2).encode()
    for tag, (a, b) in {"full": (d1, d2), "no_final_newline": (d1[:-1], d2)}.items():
        p1, p2 = str(tmp_path / (tag + "_1.fastq")), str(tmp_path / (tag + "_2.fastq"))
        open(p1, "wb").write(a)
        open(p2, "wb").write(b)
        w1, w2 = reference_rule(a), reference_rule(b)
        for block, per_item in [(64, 1), (333, 7), (4096, 100), (1 << 22, 65536)]:
            r = subprocess.run([harness, "pair", p1, p2, str(block), str(per_item)], capture_output=True, text=True, check=True)
            lines = r.stdout.splitlines()
            assert lines[0] == "%d %d" % (a.count(b"\n"), b.count(b"\n"))
            got = [tuple(int(x) for x in line.split()) for line in lines[1:]]
            n = min(len(w1), len(w2))
            assert got == [w1[i] + w2[i] for i in range(n)], (tag, block, per_item)


def test_block_chain_under_thread_sanitizer(tmp_path):
    """The only cross-thread hand-over of the block reader (lines_end / last_nl, published with release, read with
    acquire) under -fsanitize=thread: 8 threads, 4 KiB blocks, no report and the single-thread answer."""
    exe = str(tmp_path / "blocks_tsan")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", exe,
                         os.path.join(ROOT, "tests", "harness", "blocks_harness.cpp")], capture_output=True, text=True)
    if cc.returncode != 0:
        pytest.skip("ThreadSanitizer runtime not available: " + cc.stderr[-200:])
    rnd = random.Random(1)
    data = "".join("@r%d\n%s\n+\n%s\n" % (i, "A" * n, "I" * n) for i, n in ((i, rnd.choice([0, 36, 150, 400])) for i in range(8000))).encode()
    path = str(tmp_path / "t.fastq")
    open(path, "wb").write(data)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([exe, path, "4096", "8"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stderr[-2000:]
    assert [tuple(int(x) for x in line.split()) for line in r.stdout.splitlines()] == reference_rule(data)
