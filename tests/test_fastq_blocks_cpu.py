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


def run(exe, path, block, threads):
    r = subprocess.run([exe, path, str(block), str(threads)], capture_output=True, text=True, check=True)
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
