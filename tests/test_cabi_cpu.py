"""CPU-side checks of the C ABI library: it loads, exports every declared symbol,
its host-only entry points (packing, synthetic generator) work, and compute entry
points fail loudly without a GPU."""
import re
import os

import numpy as np
import pytest

import oracle as O
from trew_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "trew_hip.h")).read()
    declared = set(re.findall(r"\b(trew_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTED_SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None
    assert lib.trew_hip_abi_version() == 4


def test_pack_reads_matches_codes_table():
    reads = [b"ACGTNacgtnRYKM\r", b"T" * 33, b"", b"GATTACA" * 20]
    words, offs, lens = capi.pack_reads(reads)
    assert list(lens) == [len(r) for r in reads]
    for r, off, n in zip(reads, offs, lens):
        for i, ch in enumerate(r):
            j, b = divmod(i, 32)
            lo = (int(words[off + 3 * j]) >> b) & 1
            hi = (int(words[off + 3 * j + 1]) >> b) & 1
            nm = (int(words[off + 3 * j + 2]) >> b) & 1
            c = O.code(chr(ch))
            if c < 0:
                assert nm == 1 and lo == 0 and hi == 0
            else:
                assert nm == 0 and 2 * hi + lo == c


def test_synthetic_generator_statistics():
    n, L = 20000, 150
    buf, st, nd = capi.synth_short_ascii(20250218, 0, n, L)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    assert all(len(r) == L for r in reads)
    telo = sum(1 for r in reads if r.count(b"TTAGGG") + r.count(b"CCCTAA") >= 20)
    junc = sum(1 for r in reads if 8 <= r.count(b"TTAGGG") + r.count(b"CCCTAA") < 20)
    assert 0.007 * n < telo < 0.013 * n
    assert 0.003 * n < junc < 0.008 * n
    nfrac = sum(r.count(b"N") for r in reads) / (n * L)
    assert 2e-4 < nfrac < 9e-4
    # deterministic and index-addressable
    buf2, _, _ = capi.synth_short_ascii(20250218, 100, 50, L)
    assert buf2 == buf[100 * (L + 1): 150 * (L + 1)]


def test_compute_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.TrewHipError):
        capi.TrewHip()


def test_pack_reads_every_byte_value_all_simd_widths():
    """All 256 byte values at every alignment through the 64-, 32- and 1-base packers (codes[], kmer.cpp:14-31)."""
    import random

    rnd = random.Random(3)
    reads = [bytes(range(256)) * 2, bytes(rnd.randrange(256) for _ in range(1000))]
    reads += [bytes(rnd.choice(b"ACGTacgtNn\r") for _ in range(n)) for n in (31, 32, 33, 63, 64, 65, 95, 96, 127, 128, 129, 191, 200)]
    reads = [r.replace(b"\n", b"X") for r in reads]  # pack_reads() joins with newlines
    words, offs, lens = capi.pack_reads(reads)
    for r, off, n in zip(reads, offs, lens):
        assert n == len(r)
        for i, ch in enumerate(r):
            j, b = divmod(i, 32)
            lo = (int(words[off + 3 * j]) >> b) & 1
            hi = (int(words[off + 3 * j + 1]) >> b) & 1
            nm = (int(words[off + 3 * j + 2]) >> b) & 1
            c = O.code(chr(ch)) if ch < 128 else -1
            assert (nm, lo, hi) == ((1, 0, 0) if c < 0 else (0, c & 1, c >> 1)), (ch, i)
