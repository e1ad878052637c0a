"""host/output.cpp (process_output, check_ans_seq, final_process_output, get_score_map) against
oracle/output_oracle.py on the CPU: the product's C++ is compiled with a small stdin harness, no GPU involved.
Covers the uint32 wrap of the reference's ResultMap (kmer.h:79) at the per-file merge point."""
import os
import random
import subprocess

import pytest

import oracle as O
from oracle.output_oracle import add_totals
from trew_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("harness") / "output_harness")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "harness", "output_harness.cpp"),
                    os.path.join(ROOT, "trew_amd", "csrc", "host", "output.cpp")], check=True)
    return exe


def _run(exe, files, min_mer):
    text = []
    for name, tables in files:
        text.append("file %s" % name)
        for t, tname in enumerate(capi.TABLE_NAMES):
            for (k, w), c in tables.get(tname, {}).items():
                text.append("%d %d %d %d %d" % (t, k, w >> 64, w & (2 ** 64 - 1), c))
    r = subprocess.run([exe, str(min_mer)], input="\n".join(text) + "\n", capture_output=True, text=True, check=True)
    return r.stdout.splitlines()


def _expected(files, min_mer):
    lines, th, tl = [], {}, {}
    for name, tables in files:
        h, lo = O.fold_tables(tables, min_mer)
        lines += O.format_sections(name, h, lo)
        add_totals(th, h)
        add_totals(tl, lo)
    return lines + O.putative_trm(th, tl)


def _random_tables(rnd, kmax, big=False):
    t = {name: {} for name in capi.TABLE_NAMES}
    motifs = ["TTAGGG", "CCCTAA", "TTAGG", "AACCCT", "ACGT", "AATT", "TTTAGGG", "AT", "TTAGGGTTAGGC", "GATC" * 3,
              "TTGCATCACACCCTCGCCG", "TTTTGCCCTCATCACACCCTCGCCTCCTTCGTGCTTGCCCCCACACTGACTGACGTGCAGTCTG"]
    for name in capi.TABLE_NAMES:
        for _ in range(rnd.randint(0, 30)):
            m = rnd.choice(motifs)
            if rnd.random() < 0.5:
                m = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(3, kmax)))
            if len(m) > kmax:
                continue
            k = len(m)
            w = O.four_to_int(m)
            key = (k, O.rot_seq(w, k) if "both" not in name else min(O.rot_seq(w, k), O.rot_seq(O.revcomp(w, k), k)))
            c = rnd.choice([1, 3, 9, 10, 11, 19, 20, 500, 10 ** 6])
            if big:
                c = rnd.choice([2 ** 32 - 1, 2 ** 32, 2 ** 32 + 7, 2 ** 33 + 25, 3 * 2 ** 31 + 11, 2 ** 31, 12])
            t[name][key] = t[name].get(key, 0) + c
    return t


@pytest.mark.parametrize("seed", range(6))
def test_output_matches_oracle(harness, seed):
    rnd = random.Random(seed)
    min_mer = rnd.choice([3, 5, 6])
    files = [("/data/f%d.fastq" % i, _random_tables(rnd, rnd.choice([12, 32, 64]))) for i in range(rnd.randint(1, 3))]
    assert _run(harness, files, min_mer) == _expected(files, min_mer)


def test_counts_wrap_like_the_reference_uint32_maps(harness):
    """ResultMap is KmerSeq -> uint32_t (kmer.h:79): the thread merge and the backward fold add modulo 2^32
    (kmer.cpp:1486-1523).  The device counts in 64 bits and the host truncates once at the merge point."""
    w = O.four_to_int("TTAGGG")
    rc = O.rot_seq(O.revcomp(w, 6), 6)
    t = {name: {} for name in capi.TABLE_NAMES}
    t["forward_high"][(6, w)] = 2 ** 32 + 40      # wraps to 40
    t["backward_high"][(6, rc)] = 2 ** 32 - 15    # folded into forward[w]: (40 + 2^32 - 15) mod 2^32 = 25
    t["both_high"][(6, w)] = 2 ** 33 + 12         # wraps to 12
    out = _run(harness, [("/x.fastq", t)], 5)
    assert out == _expected([("/x.fastq", t)], 5)
    assert "6,TTAGGG,25,0,12,+" in out
    rnd = random.Random(99)
    files = [("/big%d" % i, _random_tables(rnd, 32, big=True)) for i in range(3)]
    assert _run(harness, files, 5) == _expected(files, 5)


@pytest.mark.parametrize("low_fb,high_fb", [((100, 60), (10, 50)), ((100, 60), (50, 10)), ((40, 20), (10, 20)), ((30, 30), (5, 50)),
                                             ((0, 0), (12, 40)), ((20, 10), (0, 0)), ((60, 20), (20, 60)), ((20, 60), (90, 30))])
def test_putative_trm_strand_decision(harness, low_fb, high_fb):
    """final_process_output's strand ladder (kmer.cpp:2603-2650): agreement, one-sided verdicts, opposite verdicts
    (ratio by cross multiplication, ties by total), no verdict."""
    w = O.rot_seq(O.four_to_int("TTAGGG"), 6)
    rc = O.rot_seq(O.revcomp(w, 6), 6)
    fwd_key, bwd_key = (w, rc) if w < rc else (rc, w)
    t = {name: {} for name in capi.TABLE_NAMES}
    for name, (f, b) in (("low", low_fb), ("high", high_fb)):
        if f:
            t["forward_" + name][(6, fwd_key)] = f
        if b:
            t["forward_" + name][(6, bwd_key)] = b
    other = O.rot_seq(O.four_to_int("TTTAGGG"), 7)
    t["both_high"][(7, min(other, O.rot_seq(O.revcomp(other, 7), 7)))] = 25  # keeps the section non-trivial
    files = [("/s.fastq", t)]
    assert _run(harness, files, 5) == _expected(files, 5)
