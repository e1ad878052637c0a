"""Parity of the HIP path against the CPU oracle, through the C ABI.  GPU only."""
import os
import random

import numpy as np
import pytest

import oracle as O
import trew_amd as T
from trew_amd import capi
from conftest import GOLDEN, read_fastq
from helpers import EDGE_LENGTHS, edge_reads, mixed_segments

pytestmark = pytest.mark.gpu

KAT32 = ["TTGCATCACACCCTCGCCG", "TTAGGG", "TTAGAGCCCACA", "TTTTGCCCTCATCACACCCTCGCCTCCTTCGC"]


@pytest.mark.parametrize("flags", [0, T.FLAG_NO_FILTER])
@pytest.mark.parametrize("motif", KAT32)
def test_k_mer_test_kat_on_gpu(motif, flags):
    # test.cpp:172-214 restated against the HIP path
    r = T.k_mer_check((motif * 20).encode(), 5, 32, 0.5, 0.8, flags=flags)
    assert len(r["hist_high"]) == 1
    (k, w), c = next(iter(r["hist_high"].items()))
    assert k == len(motif) == r["k_high"]
    assert min(w, O.rot_seq(O.revcomp(w, k), k)) == O.four_to_int(motif)
    assert c == len(motif) * 19 + 1
    assert r == O.segment_check(O.OracleParams(), (motif * 20).encode())


def test_survey_segment_vector_on_gpu():
    s = "TTAGGG" * 7 + "TTANGG" + "TTAGGG" + "TTAGGC" + "TTAGGG" * 2 + "TTA"
    r = T.k_mer_check(s.encode(), 5, 18)
    assert (r["k_high"], r["k_low"], r["seq_high"]) == (6, 6, 213)
    assert {O.int_to_four(w, k): c for (k, w), c in r["hist_high"].items()} == {"TTAGGG": 58, "TTAGGC": 6}


def _segment_parity(segs, min_mer, max_mer, low, high, flags):
    p = O.OracleParams(min_mer=min_mer, max_mer=max_mer, low=low, high=high)
    exp = [O.segment_check(p, s) for s in segs]
    with T.TrewHip(mode=T.MODE_SEGMENT, min_mer=min_mer, max_mer=max_mer, low=low, high=high, flags=flags,
                   max_batch_reads=len(segs) + 8, max_batch_words=1 << 20) as t:
        b = t.submit_reads(segs)
        t.wait()
        kh, kl, sh, sl = t.segment_results(len(segs))
        tabs = t.collect()
    bad = []
    for i, e in enumerate(exp):
        got = (int(kh[i]), int(kl[i]), int(sh[i]), int(sl[i]))  # sh/sl: Python ints (hi << 64 | lo)
        want = (e["k_high"], e["k_low"], e["seq_high"], e["seq_low"])
        if got != want:
            bad.append((i, segs[i], got, want))
    assert not bad, bad[:5]
    want_h, want_l = {}, {}
    for e in exp:
        for key, c in e["hist_high"].items():
            want_h[key] = want_h.get(key, 0) + c
        for key, c in e["hist_low"].items():
            want_l[key] = want_l.get(key, 0) + c
    assert tabs["forward_high"] == want_h
    assert tabs["forward_low"] == want_l
    return exp


@pytest.mark.parametrize("flags", [0, T.FLAG_NO_FILTER])
def test_segment_parity_mixed(flags):
    segs = mixed_segments(11, 1500, [20, 33, 64, 75, 76, 95, 96, 100, 149])
    exp = _segment_parity(segs, 5, 32, 0.5, 0.8, flags)
    assert sum(1 for e in exp if e["k_high"]) > 200 and sum(1 for e in exp if e["k_low"]) > 300


def test_segment_parity_long_segments():
    segs = mixed_segments(12, 300, [150, 160, 299, 320, 500, 640, 1000, 1023])
    _segment_parity(segs, 5, 32, 0.5, 0.8, 0)


@pytest.mark.parametrize("mn,mx,low,high", [(3, 12, 0.5, 0.8), (5, 32, 0.3, 0.6), (8, 20, 0.7, 0.95), (6, 6, 0.5, 0.5), (5, 32, 1.0, 1.0)])
def test_segment_parity_other_params(mn, mx, low, high):
    segs = mixed_segments(13, 600, [30, 75, 100, 150])
    _segment_parity(segs, mn, mx, low, high, 0)


KAT128 = [
    "TGCAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
    "TTAGGG",
    "TTAGAGCCCACA",
    "TTTTGCCCTCATCACACCCTCGCCTCCTTCGC",
    "TTTTGCCCTCATCACACCCTCGCCTCCTTCGTGCTTGCCCCCACACTGACTGACGTGCAGTCTG",
]


@pytest.mark.parametrize("flags", [0, T.FLAG_NO_FILTER])
@pytest.mark.parametrize("motif", KAT128)
def test_k_mer_128_test_kat_on_gpu(motif, flags):
    # test.cpp:216-258 restated against the HIP path (128-bit words)
    r = T.k_mer_check((motif * 10).encode(), 5, 64, 0.5, 0.8, flags=flags)
    assert len(r["hist_high"]) == 1
    (k, w), c = next(iter(r["hist_high"].items()))
    m = O.four_to_int(motif)
    assert k == len(motif) == r["k_high"]
    assert min(w, O.rot_seq(O.revcomp(w, k), k)) == min(m, O.rot_seq(O.revcomp(m, k), k))
    assert c == len(motif) * 9 + 1
    assert r == O.segment_check(O.OracleParams(max_mer=64), (motif * 10).encode())


def _wide_segments(seed, count, lengths):
    import random

    from helpers import mutate, periodic

    rnd = random.Random(seed)
    out = mixed_segments(seed, count // 2, lengths)
    for i in range(count - count // 2):
        n = rnd.choice(lengths)
        unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(28, 64)))
        out.append(mutate(periodic(unit, n, rnd.randint(0, 9)), rnd, p_sub=rnd.choice([0, 0.005, 0.02]), p_n=rnd.choice([0, 0, 0.004])).encode())
    return out


@pytest.mark.parametrize("mn,mx", [(5, 64), (3, 40), (33, 64), (5, 63)])
def test_segment_parity_wide_words(mn, mx):
    segs = _wide_segments(41, 700, [130, 150, 200, 256, 300, 640])
    exp = _segment_parity(segs, mn, mx, 0.5, 0.8, 0)
    assert sum(1 for e in exp if e["k_high"] > 32) > 50


def test_short_parity_wide_words():
    import random

    from helpers import mutate, periodic

    rnd = random.Random(43)
    reads = edge_reads(7)
    for n in (150, 250, 300, 600, 1000):
        for _ in range(40):
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(30, 64)))
            reads.append(mutate(periodic(unit, n, rnd.randint(0, 9)), rnd, p_sub=rnd.choice([0, 0.01])).encode())
    want = _short_parity(reads, max_mer=64)
    assert any(k > 32 for t in want.values() for (k, _) in t)
    # the bundled fixture at `3 64`: the five k = 3 rows
    fx = read_fastq(os.path.join(GOLDEN, "test.fastq"))
    w2 = _short_parity(fx, min_mer=3, max_mer=64)
    h, lo = O.fold_tables(w2, 3)
    assert "3,TTA,157,105,0,-" in O.format_sections("f", h, lo)


def test_filter_is_sound():
    """The prefilter may keep too much, never too little: every k whose exact
    MAX/COUNT reaches LOW must be a candidate."""
    segs = mixed_segments(21, 800, [40, 75, 90, 150, 300])
    p = O.OracleParams()
    with T.TrewHip(mode=T.MODE_SEGMENT, max_batch_reads=len(segs) + 8, max_batch_words=1 << 20) as t:
        b = t.host_batch(*capi.pack_reads(segs))
        cand = t.filter_masks(b, 1)
    n_pass = n_cand = 0
    for i, s in enumerate(segs):
        st = O.segment_stats(p, s, 5, 32)
        for k, (cnt, mx, _) in st.items():
            is_c = (int(cand[i, 0]) >> (k - 1)) & 1
            n_cand += is_c
            if cnt and mx / cnt >= 0.5:
                n_pass += 1
                assert is_c, (i, k, s)
    assert n_pass > 500
    # and it is selective on random sequence
    rnd = np.random.default_rng(5)
    rand = ["".join("ACGT"[x] for x in rnd.integers(0, 4, 75)).encode() for _ in range(4000)]
    with T.TrewHip(mode=T.MODE_SEGMENT, max_batch_reads=5000, max_batch_words=1 << 20) as t:
        cand = t.filter_masks(t.host_batch(*capi.pack_reads(rand)), 1)
    assert np.count_nonzero(cand) <= 4


def _short_parity(reads, **kw):
    p = O.OracleParams(**{k: v for k, v in kw.items() if k in ("min_mer", "max_mer", "low", "high")})
    want = O.run_short(p, reads)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22, **kw) as t:
        t.submit_reads(reads)
        t.wait()
        got = t.collect()
    for name in T.TABLE_NAMES:
        assert got[name] == want[name], name
    return want


def test_short_parity_edge_reads():
    want = _short_parity(edge_reads())
    assert all(len(want[n]) > 0 for n in T.TABLE_NAMES)


def test_short_parity_edge_reads_small_k():
    _short_parity(edge_reads(5), min_mer=3, max_mer=12)


def test_short_parity_synthetic_20k():
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 20000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    want = _short_parity(reads)
    assert sum(want["both_high"].values()) > 10000


def test_short_fixture_is_empty():
    reads = read_fastq(os.path.join(GOLDEN, "test.fastq"))
    want = _short_parity(reads)
    assert all(len(v) == 0 for v in want.values())


def test_device_generator_matches_host():
    n, L = 5000, 150
    buf, st, nd = capi.synth_short_ascii(7, 123, n, L)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    words, offs, lens = capi.pack_reads(reads)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=n, max_batch_words=16) as t:
        stride = 3 * ((L + 31) // 32)
        d = t.malloc(n * stride * 4)
        t.synth_short_device(7, 123, n, L, d)
        got = t.d2h(d, n * stride * 4).view(np.uint32)
        # the resident batch scans to the same tables as the host-packed one
        t.submit(t.device_uniform_batch(d, n, L))
        t.wait()
        a = t.collect()
        t.free(d)
    assert np.array_equal(got, words)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=n, max_batch_words=len(words) + 8) as t:
        t.submit(t.host_batch(words, offs, lens))
        t.wait()
        b = t.collect()
    assert a == b == O.run_short(O.OracleParams(), reads)


def test_multi_slot_and_reset():
    buf, st, nd = capi.synth_short_ascii(99, 0, 6000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    want = O.run_short(O.OracleParams(), reads)
    with T.TrewHip(mode=T.MODE_SHORT, n_slots=3, max_batch_reads=2048, max_batch_words=1 << 18) as t:
        for rep in range(2):
            for i in range(0, len(reads), 2000):
                slot = (i // 2000) % 3
                t.wait(slot)
                t.submit_reads(reads[i:i + 2000], slot)
            assert t.collect() == want
            t.reset_tables()
        assert all(len(v) == 0 for v in t.collect().values())
        t.add_rows(want)
        t.add_rows(want)
        got = t.collect()
        assert {n: {k: 2 * c for k, c in want[n].items()} for n in want} == got


def test_one_copy_batches_and_untimed_contexts():
    """The [offsets][lengths][words] host layout (one H2D copy per batch) and TREW_FLAG_NO_TIMING (no HIP events):
    what the `trew` host submits.  Many small batches back to back on two slots also exercise the exact kernel's
    reset of the worklist counters (no memset between submits)."""
    buf, st, nd = capi.synth_short_ascii(4, 0, 12000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)] + edge_reads(2)
    want = O.run_short(O.OracleParams(), reads)
    for flags in (0, T.FLAG_NO_TIMING):
        with T.TrewHip(mode=T.MODE_SHORT, n_slots=2, max_batch_reads=1024, max_batch_words=1 << 17, flags=flags) as t:
            for rep in range(2):
                for j, i in enumerate(range(0, len(reads), 700)):
                    slot = j % 2
                    t.wait(slot)
                    t.submit(t.host_batch(*capi.pack_reads(reads[i:i + 700]), contiguous=(j % 3 != 0)), slot)
                if rep == 0:
                    assert t.collect() == want
            assert t.collect() == {n: {k: 2 * c for k, c in want[n].items()} for n in want}
            if flags & T.FLAG_NO_TIMING:
                with pytest.raises(T.TrewHipError):
                    t.last_timing(0)
            else:
                a, b, nflag = t.last_timing(0)
                assert a > 0 and b > 0 and 0 < nflag <= 700


def test_filter_masks_between_submits_leaves_the_queue_counters_clean():
    """trew_hip_filter_masks runs the prefilter alone on slot 0's stream and counter block; submits before and after it
    (whose exact kernels clear each other's counter blocks, no memset in between) must be unaffected."""
    buf, st, nd = capi.synth_short_ascii(6, 0, 9000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    want = O.run_short(O.OracleParams(), reads)
    with T.TrewHip(mode=T.MODE_SHORT, n_slots=1, max_batch_reads=4096, max_batch_words=1 << 18) as t:
        for i in range(0, 9000, 3000):
            t.submit_reads(reads[i:i + 3000], 0)
            t.wait(0)
            cand = t.filter_masks(t.host_batch(*capi.pack_reads(reads[i:i + 3000])), 3)
            assert np.count_nonzero(cand.any(axis=1)) >= t.last_timing(0)[2] > 0  # the masks are per segment, flagging per read
            cand2 = t.filter_masks(t.host_batch(*capi.pack_reads(reads[:500])), 3)  # twice in a row
            assert cand2.shape == (500, 3)
        assert t.collect() == want


def test_device_pack_matches_host_pack():
    """trew_hip_submit_ascii: codes[] (kmer.cpp:14-31) applied by a kernel.  The packed words must equal trew_pack_reads word
    for word -- every byte value, either case, reads that start at every byte alignment and end at every position of a
    triple, empty reads -- and the tables of a text batch must equal those of the host-packed batch."""
    import random

    rnd = random.Random(12)
    reads = [bytes(range(256)), bytes(range(255, -1, -1)), b"", b"A", b"acgtnACGTN" * 13, b"N" * 33, b"T" * 32, b"G" * 31]
    for n in list(range(1, 70)) + [95, 96, 97, 127, 128, 129, 150, 151, 250, 999, 1000]:
        reads.append("".join(rnd.choice("ACGTacgtNn.*\r") if rnd.random() < 0.1 else rnd.choice("ACGT") for _ in range(n)).encode())
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 3000, 150)
    synth = [buf[s:e + 1] for s, e in zip(st, nd)]
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=1 << 14, max_batch_words=1 << 20, max_batch_ascii_bytes=1 << 22) as t:
        for contiguous in (True, False):
            got = t.pack_ascii(t.ascii_batch(reads, contiguous=contiguous))
            want, _, _ = capi.pack_reads(reads)
            assert got.shape == want.shape and (got == want).all(), np.nonzero(got != want)[0][:10]
        got = t.pack_ascii(t.ascii_batch(synth, uniform=150))
        want, _, _ = capi.pack_reads(synth)
        assert (got == want).all()
        # tables: text batch (ragged and uniform) == host-packed batch == oracle
        short = [r for r in reads if len(r) <= 1000] + synth
        want_t = O.run_short(O.OracleParams(), short)
        t.submit_ascii(t.ascii_batch(short))
        t.wait()
        assert t.collect() == want_t
        t.reset_tables()
        t.submit_ascii(t.ascii_batch(synth, uniform=150), 1)
        t.wait(1)
        assert t.collect() == O.run_short(O.OracleParams(), synth)
        with pytest.raises(T.TrewHipError):
            t.submit_ascii(t.ascii_batch([b"A" * 1001]))
    with pytest.raises(T.TrewHipError, match="max_batch_ascii_bytes"):
        with T.TrewHip(mode=T.MODE_SHORT) as t:
            t.submit_ascii(t.ascii_batch([b"ACGT" * 10]))
    # pair mode: reads 2i, 2i+1 are mates
    b1, b2, st, nd = capi.synth_pair_ascii(20250218, 0, 4000, 150)
    inter = []
    for s_, e_ in zip(st, nd):
        inter += [b1[s_:e_ + 1], b2[s_:e_ + 1]]
    with T.TrewHip(mode=T.MODE_PAIR, max_batch_reads=1 << 14, max_batch_words=1 << 20, max_batch_ascii_bytes=1 << 22) as t:
        t.submit_ascii(t.ascii_batch(inter))
        t.wait()
        assert t.collect() == O.run_pair(O.OracleParams(), inter[0::2], inter[1::2])


def test_empty_and_tiny_batches():
    with T.TrewHip(mode=T.MODE_SHORT) as t:
        t.submit_reads([])
        t.wait()
        t.submit_reads([b"", b"ACG", b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN"])
        t.wait()
        assert all(len(v) == 0 for v in t.collect().values())
    with pytest.raises(T.TrewHipError):
        with T.TrewHip(mode=T.MODE_SHORT) as t:
            t.submit_reads([b"A" * 1001])
    # a batch whose reads point outside its words is refused on the host: the kernels index words[] with these numbers
    words, offs, lens = capi.pack_reads([b"ACGT" * 30, b"TTAGGG" * 20])
    for bad_offs, bad_lens in ((offs + np.uint32(len(words)), lens), (offs, lens + np.uint32(4000))):
        with T.TrewHip(mode=T.MODE_LONG if bad_lens is not lens else T.MODE_SHORT) as t:
            with pytest.raises(T.TrewHipError):
                t.submit(t.host_batch(words, bad_offs, bad_lens))
    with T.TrewHip(mode=T.MODE_SHORT) as t:
        w = np.zeros(60, dtype=np.uint32)
        with pytest.raises(T.TrewHipError):
            t.submit(capi.Batch(w.ctypes.data, len(w), None, None, 150, 15, 5, 0, 150))  # 5 reads x 15 words > 60 words


# ---------------------------------------------------------------- paired-end (buffer_task_pair)
def _pair_parity(r1, r2, **kw):
    p = O.OracleParams(**{k: v for k, v in kw.items() if k in ("min_mer", "max_mer", "low", "high")})
    want = O.run_pair(p, r1, r2)
    reads = []
    for a, b in zip(r1, r2):
        reads += [a, b]
    with T.TrewHip(mode=T.MODE_PAIR, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22, **kw) as t:
        t.submit_reads(reads)
        t.wait()
        got = t.collect()
    for name in T.TABLE_NAMES:
        assert got[name] == want[name], name
    return want


def test_pair_parity_synthetic_2x150():
    b1, b2, st, nd = capi.synth_pair_ascii(20250218, 0, 20000, 150)
    r1 = [b1[s:e + 1] for s, e in zip(st, nd)]
    r2 = [b2[s:e + 1] for s, e in zip(st, nd)]
    want = _pair_parity(r1, r2)
    assert sum(want["both_high"].values()) > 10000 and len(want["forward_high"]) > 0


def _revcomp(s):
    return s.translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1]


def test_pair_parity_edge_pairs():
    import random

    from helpers import mutate, periodic

    rnd = random.Random(17)
    r1, r2 = [], []
    units = ["TTAGGG", "CCCTAA", "AT", "TTAGG", "TTTAGGG", "ACGTACGTAC", "TTGCATCACACCCTCGCCG", "AATT", "A"]
    for n1, n2 in [(150, 150), (100, 100), (60, 60), (150, 100), (101, 151), (127, 128), (128, 127), (40, 40), (20, 21),
                   (12, 30), (9, 9), (250, 250), (300, 90)]:
        for u in units:
            frag = periodic(u, n1 + n2, rnd.randint(0, 5))
            frag = mutate(frag, rnd, p_sub=rnd.choice([0.0, 0.01, 0.04]), p_n=rnd.choice([0.0, 0.0, 0.01]))
            a = frag[:n1].encode()
            b = _revcomp(frag[n1:].encode())
            r1.append(a)
            r2.append(b)
            # only one mate repetitive / repeat on part of the fragment
            rand = "".join(rnd.choice("ACGT") for _ in range(n2)).encode()
            r1.append(a)
            r2.append(rand)
            r1.append(_revcomp(rand)[:n1].ljust(n1, b"A")[:n1])
            r2.append(b)
            h = (n1 + n2) // 3
            frag2 = (periodic(u, h) + "".join(rnd.choice("ACGT") for _ in range(n1 + n2 - h)))
            r1.append(frag2[:n1].encode())
            r2.append(_revcomp(frag2[n1:].encode()))
            # different motifs on the two mates
            r1.append(periodic(u, n1).encode())
            r2.append(periodic("GGGTTA", n2).encode())
    want = _pair_parity(r1, r2)
    assert all(len(want[n]) > 0 for n in T.TABLE_NAMES)
    _pair_parity(r1, r2, min_mer=3, max_mer=12)


def _g1_pairs(seed, count):
    """Pairs that make the whole-read block of buffer_task_pair record something (a period too long for a quarter of the read:
    the halves find nothing, the whole read does), between pairs of every kind that can follow them: fully chained telomeric
    pairs, random pairs, pairs shorter than 4 * MIN_MER (their four-segment block is skipped and the stale map survives them),
    pairs below 2 * MIN_MER (skipped altogether), mates of unequal length, an N here and there."""
    import random

    from helpers import mutate, periodic

    rnd = random.Random(seed)
    r1, r2 = [], []

    def rand(n):
        return "".join(rnd.choice("ACGT") for _ in range(n))

    for _ in range(count):
        kind = rnd.random()
        n1 = rnd.choice([100, 100, 100, 90, 120, 127, 64, 110])
        n2 = rnd.choice([n1, n1, 100, 110])
        if kind < 0.35:  # whole-read-only repeat on one or both mates
            k = rnd.randint(min(n1, n2) // 4 + 1, min(32, min(n1, n2) // 2))
            unit = rand(k)
            a = mutate(periodic(unit, n1, rnd.randint(0, k - 1)), rnd, p_sub=rnd.choice([0.0, 0.0, 0.02]), p_n=rnd.choice([0.0, 0.0, 0.01]))
            b = _revcomp(periodic(unit, n2, rnd.randint(0, k - 1)).encode()).decode() if rnd.random() < 0.5 else rand(n2)
        elif kind < 0.5:  # fully chained telomeric pair
            frag = periodic(rnd.choice(["TTAGGG", "CCCTAA", "TTTAGGG"]), n1 + n2, rnd.randint(0, 5))
            a, b = frag[:n1], _revcomp(frag[n1:].encode()).decode()
        elif kind < 0.6:  # telomeric on the first mate only
            a, b = periodic("TTAGGG", n1, rnd.randint(0, 5)), rand(n2)
        elif kind < 0.75:  # shorter than 4 * MIN_MER: no four-segment block
            m = rnd.randint(10, 19)
            u = rand(rnd.randint(5, m // 2))
            a, b = periodic(u, m + rnd.randint(0, 3)), (periodic(u, m) if rnd.random() < 0.5 else _revcomp(periodic(u, m).encode()).decode())
        elif kind < 0.8:  # below 2 * MIN_MER: skipped
            a, b = rand(rnd.randint(1, 9)), rand(rnd.randint(1, 30))
        else:
            a, b = rand(n1), rand(n2)
        r1.append(a.encode())
        r2.append(b.encode())
    return r1, r2


def test_pair_compat_g1_reproduces_the_uncleared_map():
    """TREW_FLAG_COMPAT_G1: the reference's 64-bit pair branch as written (kmer.cpp:467-505 has no clear of temp_result_left;
    its 128-bit twin has, 722-723) for one consumer thread -- against the oracle run with compat_g1 on the same pairs in the same
    order, as one batch and cut into batches at every size (rows of a batch's last pairs reach the next batch through the carry),
    for two k ranges; and the flag changes the tables (the test has teeth), while a context without it still gives the cleared
    semantics."""
    r1, r2 = _g1_pairs(5, 700)
    for kw in ({}, {"min_mer": 3, "max_mer": 12}):
        p = O.OracleParams(compat_g1=True, **kw)
        want = O.run_pair(p, r1, r2)
        cleared = O.run_pair(O.OracleParams(**kw), r1, r2)
        assert want != cleared and sum(want["forward_high"].values()) > sum(cleared["forward_high"].values())
        assert sum(want["both_low"].values()) > sum(cleared["both_low"].values())  # stale rows that met a fully chained pair
        reads = [x for ab in zip(r1, r2) for x in ab]
        for per_batch in (len(r1), 1, 7, 64):
            with T.TrewHip(mode=T.MODE_PAIR, n_slots=1, max_batch_reads=2 * len(r1) + 8, max_batch_words=1 << 21, flags=T.FLAG_COMPAT_G1, **kw) as t:
                for at in range(0, len(r1), per_batch):
                    t.submit_reads(reads[2 * at:2 * (at + per_batch)])
                    t.wait()
                got = t.collect()
                for name in T.TABLE_NAMES:
                    assert got[name] == want[name], (kw, per_batch, name)
                # a second input on the same context starts with an empty map
                t.reset_tables()
                t.submit_reads(reads)
                t.wait()
                assert t.collect() == want
        with T.TrewHip(mode=T.MODE_PAIR, n_slots=1, max_batch_reads=2 * len(r1) + 8, max_batch_words=1 << 21, **kw) as t:
            t.submit_reads(reads)
            t.wait()
            assert t.collect() == cleared
    # the flag is refused where it cannot mean anything
    for bad in (dict(mode=T.MODE_SHORT, n_slots=1), dict(mode=T.MODE_PAIR, n_slots=2), dict(mode=T.MODE_PAIR, n_slots=1, max_mer=40)):
        with pytest.raises(T.TrewHipError, match="COMPAT_G1"):
            T.TrewHip(flags=T.FLAG_COMPAT_G1, **bad)


# ---------------------------------------------------------------- long reads (buffer_task_long)
def _long_reads(seed, count):
    import random

    from helpers import mutate, periodic

    rnd = random.Random(seed)
    out = []
    for i in range(count):
        n = rnd.choice([100, 149, 150, 151, 299, 300, 301, 449, 450, 600, 1000, 1499, 2300, 5000, 12000])
        body = "".join(rnd.choice("ACGT") for _ in range(n))
        kind = rnd.random()
        unit = rnd.choice(["TTAGGG", "CCCTAA", "TTAGGG", "TTTAGGG", "AT", "TTAGGGTTAGGC", "ACG"])
        if kind < 0.3:
            t = min(n, rnd.choice([150, 300, 450, 700, 2000]) + rnd.randint(-40, 40))
            body = body[: n - t] + mutate(periodic(unit, t, rnd.randint(0, 5)), rnd, p_sub=rnd.choice([0.0, 0.02, 0.05]))
        elif kind < 0.55:
            t = min(n, rnd.choice([150, 300, 450, 700, 2000]) + rnd.randint(-40, 40))
            body = mutate(periodic(unit, t, rnd.randint(0, 5)), rnd, p_sub=rnd.choice([0.0, 0.02, 0.05])) + body[t:]
        elif kind < 0.7:
            body = mutate(periodic(unit, n, rnd.randint(0, 5)), rnd, p_sub=rnd.choice([0.0, 0.03]), p_n=rnd.choice([0, 0.002]))
        elif kind < 0.8:
            # repeat at both ends with different motifs, or a switch of motif inside the repeat
            t = min(n // 2, 400)
            body = periodic(unit, t) + body[t: n - t] + periodic("GGGTTA" if rnd.random() < 0.5 else "TTAGG", t)
        out.append(body[:n].encode())
    return out


@pytest.mark.parametrize("slice_length", [150, 100])
def test_long_parity(slice_length):
    reads = _long_reads(31, 400)
    p = O.OracleParams(slice_len=slice_length)
    want = O.run_long(p, reads)
    kept = [r for r in reads if len(r) >= slice_length]
    with T.TrewHip(mode=T.MODE_LONG, slice_length=slice_length, max_batch_reads=len(kept) + 8, max_batch_words=1 << 22) as t:
        t.submit_reads(kept)
        t.wait()
        got = t.collect()
    for name in T.TABLE_NAMES:
        assert got[name] == want[name], name
    assert all(len(want[n]) > 0 for n in T.TABLE_NAMES)


def test_long_fixture_is_empty():
    reads = read_fastq(os.path.join(GOLDEN, "test_long.fastq"))
    with T.TrewHip(mode=T.MODE_LONG, max_batch_reads=64, max_batch_words=1 << 20) as t:
        t.submit_reads(reads)
        t.wait()
        got = t.collect()
    assert all(len(v) == 0 for v in got.values())
    assert got == O.run_long(O.OracleParams(), reads)


# ---------------------------------------------------------------- randomized cross-checks, every mode
def _fuzz_reads(rnd, count, maxlen):
    from helpers import mutate, periodic

    out = []
    for _ in range(count):
        n = rnd.choice([rnd.randint(1, 60), rnd.randint(60, 260), rnd.randint(1, maxlen)])
        kind = rnd.random()
        if kind < 0.2:
            s = "".join(rnd.choice("ACGT") for _ in range(n))
        elif kind < 0.3:
            s = "".join(rnd.choice("AT") for _ in range(n))  # low complexity: many runs, many classes
        else:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 70)))
            s = periodic(unit, n, rnd.randint(0, 11))
            s = mutate(s, rnd, p_sub=rnd.choice([0, 0.01, 0.05, 0.2]), p_n=rnd.choice([0, 0, 0.01, 0.1]))
            if rnd.random() < 0.3:  # junction with a second repeat or random tail
                cut = rnd.randint(0, n)
                u2 = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(2, 40)))
                s = s[:cut] + periodic(u2, n - cut)
        if rnd.random() < 0.05:
            s = s.lower()
        out.append(s[:n].encode())
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("TREW_FUZZ_SEEDS", "40"))))
def test_fuzz_all_modes(seed):
    import random

    rnd = random.Random(1000 + seed)
    mn = rnd.choice([3, 4, 5, 6, 9, 17])
    mx = rnd.choice([mn, mn + 1, 12, 20, 31, 32, 33, 48, 63, 64])
    mx = max(mn, mx)
    low = rnd.choice([0.5, 0.3, 0.51, 2 / 3, 0.75, 1.0])
    high = max(low, rnd.choice([0.8, 0.6, 0.9, 1.0]))
    kw = dict(min_mer=mn, max_mer=mx, low=low, high=high)
    p = O.OracleParams(**kw)
    # short
    reads = _fuzz_reads(rnd, 250, 1000)
    want = O.run_short(p, reads)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22, **kw) as t:
        t.submit_reads(reads)
        t.wait()
        assert t.collect() == want, ("short", kw)
    # pair
    r1 = _fuzz_reads(rnd, 150, 400)
    r2 = [_revcomp(r) if rnd.random() < 0.6 else x for r, x in zip(r1, _fuzz_reads(rnd, 150, 400))]
    want = O.run_pair(p, r1, r2)
    both = [x for pr in zip(r1, r2) for x in pr]
    with T.TrewHip(mode=T.MODE_PAIR, max_batch_reads=len(both) + 8, max_batch_words=1 << 22, **kw) as t:
        t.submit_reads(both)
        t.wait()
        assert t.collect() == want, ("pair", kw)
    # long
    sl = rnd.choice([2 * mx, 150, 200, 2 * mx + 7])
    sl = max(sl, 2 * mx)
    pl = O.OracleParams(slice_len=sl, **kw)
    lr = [r for r in _fuzz_reads(rnd, 120, 4000) if len(r) >= sl]
    want = O.run_long(pl, lr)
    with T.TrewHip(mode=T.MODE_LONG, slice_length=sl, max_batch_reads=len(lr) + 8, max_batch_words=1 << 22, **kw) as t:
        t.submit_reads(lr)
        t.wait()
        assert t.collect() == want, ("long", kw, sl)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TREW_GROUP_SEEDS", "12"))))
def test_group_pass_matches_oracle_and_wave_per_segment(seed):
    """The exact kernel's group pass (four segments in lock step, 16 lanes each: decide_group, row-space routing, k_mer_target as
    a whole-read row count) only runs on batches whose halves fit 3- or 5-word masks, which the fuzz above (reads up to 1000
    bases) never is.  Ragged batches of short reads -- noisy repeats, junctions, N, low-complexity words, every baseline
    combination, k ranges of 1 .. 32 values -- against the oracle, and against the same library with every segment decided by
    a wave of its own (TREW_FLAG_DEBUG_NO_GROUP)."""
    import random

    from helpers import mutate, periodic

    rnd = random.Random(4100 + seed)
    mn = rnd.choice([3, 4, 5, 5, 6, 9])
    mx = max(mn, rnd.choice([mn, 12, 20, 31, 32, 32, mn + 31 if mn + 31 <= 32 else 32]))
    low = rnd.choice([0.5, 0.5, 0.3, 0.51, 2 / 3, 0.75, 1.0])
    high = max(low, rnd.choice([0.8, 0.6, 0.9, 0.9, 1.0]))
    kw = dict(min_mer=mn, max_mer=mx, low=low, high=high)
    maxlen = rnd.choice([150, 151, 190, 250, 318])
    reads = []
    for _ in range(1500):
        n = rnd.choice([maxlen, maxlen, rnd.randint(4 * mn, maxlen), rnd.randint(1, maxlen)])
        kind = rnd.random()
        if kind < 0.1:
            s = "".join(rnd.choice("ACGT") for _ in range(n))
        elif kind < 0.17:
            s = "".join(rnd.choice("AT") for _ in range(n))
        else:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([rnd.randint(1, 12), rnd.randint(1, 40), 6])))
            s = mutate(periodic(unit, n, rnd.randint(0, 11)), rnd, p_sub=rnd.choice([0, 0.01, 0.01, 0.03, 0.08, 0.2]), p_n=rnd.choice([0, 0, 0, 0.005, 0.02, 0.1]))
            r = rnd.random()
            if r < 0.25:  # junction at the middle or anywhere
                cut = rnd.choice([n // 2, rnd.randint(0, n)])
                tail = "".join(rnd.choice("ACGT") for _ in range(n - cut)) if rnd.random() < 0.5 else periodic("".join(rnd.choice("ACGT") for _ in range(rnd.randint(2, 20))), n - cut)
                s = s[:cut] + tail if rnd.random() < 0.5 else tail + s[:cut]
        reads.append(s[:n].encode())
    want = O.run_short(O.OracleParams(**kw), reads)
    got = {}
    for flags in (0, T.FLAG_DEBUG_NO_GROUP):
        with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22, flags=flags, **kw) as t:
            t.reset_tables()  # the fall-back counters are per device: start from zero
            t.submit_reads(reads)
            t.wait()
            got[flags] = t.collect()
            c = t.debug_counters()
            if flags:
                assert c["group_punt"] == c["group_routed"] == c["group_target"] == 0, c  # the flag really turns the pass off
    assert got[0] == want, ("group pass", kw, maxlen, _table_diff(got[0], want))
    assert got[T.FLAG_DEBUG_NO_GROUP] == want, ("wave per segment", kw, maxlen)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TREW_GROUP_SEEDS", "12"))))
def test_group_pass_pairs_match_oracle_and_wave_per_segment(seed):
    """The pair driver's group pass (the four half-read segments of a pair decided in lock step; a fully chained pair recorded
    straight from the rows) on ragged batches of short pairs -- fragments that are repeats from end to end, repeats in one mate or
    one half only, mates of unequal length, N, noise -- against the oracle and against TREW_FLAG_DEBUG_NO_GROUP."""
    import random

    from helpers import mutate, periodic

    rnd = random.Random(9100 + seed)
    mn = rnd.choice([3, 4, 5, 5, 6, 9])
    mx = max(mn, rnd.choice([mn, 12, 20, 31, 32, 32]))
    low = rnd.choice([0.5, 0.5, 0.3, 0.51, 2 / 3, 0.75, 1.0])
    high = max(low, rnd.choice([0.8, 0.6, 0.9, 0.9, 1.0]))
    kw = dict(min_mer=mn, max_mer=mx, low=low, high=high)
    maxlen = rnd.choice([150, 151, 200, 250, 300])
    r1, r2 = [], []
    for _ in range(900):
        n1 = rnd.choice([maxlen, maxlen, rnd.randint(4 * mn, maxlen), rnd.randint(1, maxlen)])
        n2 = n1 if rnd.random() < 0.7 else rnd.choice([maxlen, rnd.randint(1, maxlen)])
        unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([rnd.randint(1, 12), rnd.randint(1, 40), 6])))
        frag = mutate(periodic(unit, n1 + n2, rnd.randint(0, 11)), rnd, p_sub=rnd.choice([0, 0.01, 0.01, 0.03, 0.08]), p_n=rnd.choice([0, 0, 0, 0.005, 0.02]))
        kind = rnd.random()
        if kind < 0.15:
            frag = "".join(rnd.choice("ACGT") for _ in range(n1 + n2))
        elif kind < 0.4:  # repeat in part of the fragment only
            cut = rnd.choice([n1 // 2, n1, n1 + n2 // 2, rnd.randint(0, n1 + n2)])
            rest = "".join(rnd.choice("ACGT") for _ in range(n1 + n2 - cut))
            frag = frag[:cut] + rest if rnd.random() < 0.5 else rest + frag[:cut]
        a = frag[:n1].encode()
        b = frag[n1:n1 + n2].encode()
        r1.append(a)
        r2.append(_revcomp(b) if rnd.random() < 0.7 else b)
    want = O.run_pair(O.OracleParams(**kw), r1, r2)
    both = [x for pr in zip(r1, r2) for x in pr]
    for flags in (0, T.FLAG_DEBUG_NO_GROUP):
        with T.TrewHip(mode=T.MODE_PAIR, max_batch_reads=len(both) + 8, max_batch_words=1 << 22, flags=flags, **kw) as t:
            t.submit_reads(both)
            t.wait()
            got = t.collect()
        assert got == want, ("group pass" if not flags else "wave per segment", kw, maxlen, _table_diff(got, want))


@pytest.mark.parametrize("seed", range(int(os.environ.get("TREW_GROUP_SEEDS", "12"))))
def test_group_pass_long_reads_match_oracle_and_wave_per_slice(seed):
    """The long driver with two reads per wave (run_long_groups: 32-lane groups, a slice per trip, records from registers, the
    wave-per-slice code for the middle slice / groups that give up / reads that chain from end to end): reads with repeat tails
    at either or both ends, reads that are one repeat from end to end (the forward chain runs through: forward -> both), noisy
    ONT-like repeats, N, slice lengths up to 159 -- against the oracle and against TREW_FLAG_DEBUG_NO_GROUP."""
    import random

    from helpers import mutate, periodic

    rnd = random.Random(5300 + seed)
    mn = rnd.choice([3, 5, 5, 6, 9])
    mx = max(mn, rnd.choice([12, 20, 31, 32, 32]))
    sl = max(2 * mx, rnd.choice([64, 100, 128, 150, 150, 159]))
    low = rnd.choice([0.5, 0.5, 0.3, 2 / 3, 0.75])
    high = max(low, rnd.choice([0.8, 0.9, 0.9, 1.0]))
    kw = dict(min_mer=mn, max_mer=mx, low=low, high=high)
    reads = []
    for _ in range(260):
        n = rnd.choice([rnd.randint(sl, 3 * sl), rnd.randint(sl, 2500), rnd.randint(2000, 7000)])
        body = "".join(rnd.choice("ACGT") for _ in range(n))
        unit = rnd.choice(["TTAGGG", "CCCTAA", "TTAGGG", "TTTAGGG", "AT", "TTAGGGTTAGGC", "ACG", "".join(rnd.choice("ACGT") for _ in range(rnd.randint(2, 30)))])
        psub = rnd.choice([0.0, 0.01, 0.03, 0.05, 0.05, 0.1])
        pn = rnd.choice([0, 0, 0, 0.002, 0.01])
        kind = rnd.random()
        if kind < 0.3:  # 3' tail
            t = min(n, rnd.randint(sl // 2, 3000))
            body = body[: n - t] + mutate(periodic(unit, t, rnd.randint(0, 5)), rnd, p_sub=psub, p_n=pn)
        elif kind < 0.55:  # 5' tail
            t = min(n, rnd.randint(sl // 2, 3000))
            body = mutate(periodic(unit, t, rnd.randint(0, 5)), rnd, p_sub=psub, p_n=pn) + body[t:]
        elif kind < 0.75:  # one repeat from end to end
            body = mutate(periodic(unit, n, rnd.randint(0, 5)), rnd, p_sub=rnd.choice([0.0, 0.01, 0.03]), p_n=rnd.choice([0, 0, 0.001]))
        elif kind < 0.9:  # both ends, the same or different motifs
            t = min(n // 2, rnd.randint(sl, 1500))
            u2 = unit if rnd.random() < 0.5 else rnd.choice(["GGGTTA", "TTAGG", "CCCTAA"])
            body = mutate(periodic(unit, t), rnd, p_sub=psub) + body[t: n - t] + mutate(periodic(u2, t), rnd, p_sub=psub)
        reads.append(body[:n].encode())
    want = O.run_long(O.OracleParams(slice_len=sl, **kw), reads)
    assert sum(len(v) for v in want.values()) > 0
    for flags in (0, T.FLAG_DEBUG_NO_GROUP):
        with T.TrewHip(mode=T.MODE_LONG, slice_length=sl, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22, flags=flags, **kw) as t:
            t.submit_reads(reads)
            t.wait()
            got = t.collect()
        assert got == want, ("two reads per wave" if not flags else "wave per slice", kw, sl, _table_diff(got, want))


def test_group_pass_takes_almost_every_read_of_the_bench_workload():
    """On the synthetic reads of config 2 the group pass hands back under 2 % of the flagged reads (round 4: 0.8 %) -- N, noisy
    repeats and junction reads included -- and the tables equal the oracle's."""
    buf, st, nd = capi.synth_short_ascii(20250218, 3_000_000, 200000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    want = O.run_short(O.OracleParams(), reads)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 23) as t:
        t.reset_tables()
        t.submit_reads(reads)
        t.wait()
        assert t.collect() == want
        c = t.debug_counters()
        flagged = t.last_timing(0)[2]
        assert flagged > 3000 and c["group_routed"] + c["group_target"] < 0.02 * flagged, (c, flagged)


def test_regressions_found_by_fuzzing():
    # (1) LDS mask words: a 176-base segment uses the NW=10 kernels but only 3 mask words; the run-based
    #     path once wrote past them and cleared the N mask of the first bases (leading N became valid)
    seg = b"NCGTACCGACT" + b"AGGG" * 41 + b"A"
    for mn, mx in ((17, 48), (5, 32)):
        assert T.k_mer_check(seg, mn, mx, 0.3, 0.9) == O.segment_check(O.OracleParams(min_mer=mn, max_mer=mx, low=0.3, high=0.9), seg)
    # (2) long mode: the middle slice (SLICE_LENGTH + remainder) is longer than the first/last slices
    #     the prefilter looks at; the kernels must be sized for it
    unit = "TTCCAAGTATCCTGTTATCGAGTGAA"
    read = ("ACGT" * 50 + (unit * 30)[:576]).encode()
    p = O.OracleParams(min_mer=9, max_mer=31, low=1.0, high=1.0, slice_len=200)
    with T.TrewHip(mode=T.MODE_LONG, slice_length=200, min_mer=9, max_mer=31, low=1.0, high=1.0, max_batch_reads=8, max_batch_words=1 << 16) as t:
        t.submit_reads([read])
        t.wait()
        got = t.collect()
    want = O.run_long(p, [read])
    assert got == want and sum(len(v) for v in want.values()) > 0


@pytest.mark.parametrize("read_len,mn,mx", [(150, 5, 32), (151, 5, 32), (100, 5, 32), (127, 5, 32), (128, 5, 32), (60, 3, 20),
                                             (250, 5, 32), (300, 4, 64), (75, 5, 32), (36, 5, 12), (1000, 5, 32), (97, 6, 40)])
def test_device_resident_uniform_batches(read_len, mn, mx):
    """Device-generated, device-resident uniform batches (the bench layout) at odd geometries:
    whole-read segment present/absent, halves of 18..500 bases, narrow and wide words."""
    n = 3000 if read_len <= 300 else 600
    buf, st, nd = capi.synth_short_ascii(77 + read_len, 5, n, read_len)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    want = O.run_short(O.OracleParams(min_mer=mn, max_mer=mx), reads)
    with T.TrewHip(mode=T.MODE_SHORT, min_mer=mn, max_mer=mx, max_batch_reads=n, max_batch_words=16, n_slots=2) as t:
        stride = 3 * ((read_len + 31) // 32)
        d = t.malloc(n * stride * 4 + 64)
        t.synth_short_device(77 + read_len, 5, n, read_len, d)
        half = n // 2
        # two sub-batches on two slots/streams, second one offset into the buffer
        t.submit(t.device_uniform_batch(d, half, read_len), 0)
        t.submit(t.device_uniform_batch(d + half * stride * 4, n - half, read_len), 1)
        t.wait(0)
        t.wait(1)
        got = t.collect()
        t.free(d)
    assert got == want
    assert sum(len(v) for v in want.values()) > 0


def test_tiny_table_spills_but_stays_exact():
    """A 4096-slot table (8 slots per partition) cannot hold the histograms of these reads: rows that
    find their partition full go to the spill log and are merged by collect -- results stay exact."""
    reads = edge_reads(11) + [s for s in mixed_segments(5, 1500, [150, 200, 300])]
    want = O.run_short(O.OracleParams(), reads)
    assert sum(len(v) for v in want.values()) > 4096  # more keys than slots: spilling is certain
    with T.TrewHip(mode=T.MODE_SHORT, table_log2_slots=12, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22) as t:
        for rep in range(2):  # second pass doubles every count, also through the spill path
            t.submit_reads(reads)
            t.wait()
        got = t.collect()
    assert got == {n: {k: 2 * c for k, c in want[n].items()} for n in want}
    # and with 128-bit words
    want = O.run_short(O.OracleParams(max_mer=64), reads)
    with T.TrewHip(mode=T.MODE_SHORT, max_mer=64, table_log2_slots=12, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22) as t:
        t.submit_reads(reads)
        t.wait()
        assert t.collect() == want


def _table_diff(got, want, limit=12):
    """compact description of where two table sets differ (for assertion messages)"""
    out = []
    for name in want:
        g, w = got.get(name, {}), want[name]
        for key in sorted(set(g) | set(w)):
            if g.get(key, 0) != w.get(key, 0):
                out.append("%s k=%d %s: got %d want %d" % (name, key[0], O.int_to_four(key[1], key[0]), g.get(key, 0), w.get(key, 0)))
    return "%d differing rows: %s" % (len(out), "; ".join(out[:limit]))


def _tables_sum(a, b):
    out = {n: dict(a[n]) for n in a}
    for n in b:
        for key, c in b[n].items():
            out[n][key] = out[n].get(key, 0) + c
    return out


def test_full_size_properties_config2():
    """BASELINE config 2 at full size (10 M x 150 bp, device-generated): properties that do not need
    the oracle at that size --
      * sharding invariance: one 10 M batch == the sum of four 2.5 M batches (other batch borders,
        other worklists, other queue interleaving);
      * idempotence: a second pass doubles every count exactly;
      * the uniform-geometry fast filter == the general filter (same reads as a ragged batch with
        explicit device offsets/lengths) == no filter at all, on a 2 M prefix;
      * the first 200 k reads are bit-exact against the CPU oracle."""
    import ctypes as C
    n, L = 10_000_000, 150
    stride = 3 * ((L + 31) // 32)
    seed = 20250218
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=n, max_batch_words=16, table_log2_slots=20) as t:
        d = t.malloc(n * stride * 4 + 64)
        t.synth_short_device(seed, 0, n, L, d)
        t.submit(t.device_uniform_batch(d, n, L))
        t.wait()
        whole = t.collect()
        assert sum(sum(v.values()) for v in whole.values()) > 10_000_000  # ~1.5 % repeat reads, ~70-145 windows each
        t.submit(t.device_uniform_batch(d, n, L))
        t.wait()
        assert t.collect() == {name: {k: 2 * c for k, c in whole[name].items()} for name in whole}
        t.reset_tables()
        q = n // 4
        for i in range(4):
            t.submit(t.device_uniform_batch(d + i * q * stride * 4, q, L))
            t.wait()
        assert t.collect() == whole
        # 2 M prefix: fast filter vs general filter (ragged view of the same words) vs no filter
        m = 2_000_000
        t.reset_tables()
        t.submit(t.device_uniform_batch(d, m, L))
        t.wait()
        fast = t.collect()
        offs = (np.arange(m, dtype=np.uint32) * np.uint32(stride))
        lens = np.full(m, L, dtype=np.uint32)
        d_offs, d_lens = t.malloc(m * 4), t.malloc(m * 4)
        t._chk(t.lib.trew_hip_memcpy_h2d(t.ctx, d_offs, offs.ctypes.data, m * 4), "h2d")
        t._chk(t.lib.trew_hip_memcpy_h2d(t.ctx, d_lens, lens.ctypes.data, m * 4), "h2d")
        t.reset_tables()
        t.submit(capi.Batch(d, m * stride, d_offs, d_lens, 0, 0, m, 1, L))
        t.wait()
        assert t.collect() == fast
        t.free(d_offs)
        t.free(d_lens)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=m, max_batch_words=16, table_log2_slots=20, flags=T.FLAG_NO_FILTER) as t2:
        d2 = t2.malloc(m * stride * 4 + 64)
        t2.synth_short_device(seed, 0, m, L, d2)
        t2.submit(t2.device_uniform_batch(d2, m, L))
        t2.wait()
        assert t2.collect() == fast
        t2.free(d2)
    k = 200_000
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=k, max_batch_words=16, table_log2_slots=20) as t3:
        d3 = t3.malloc(k * stride * 4 + 64)
        t3.synth_short_device(seed, 0, k, L, d3)
        t3.submit(t3.device_uniform_batch(d3, k, L))
        t3.wait()
        got = t3.collect()
        t3.free(d3)
    buf, st, nd = capi.synth_short_ascii(seed, 0, k, L)
    want, _ = O.run_short_mt_timed(O.OracleParams(), buf, st, nd, os.cpu_count() or 8)
    assert got == want


def test_full_size_config5_share():
    """BASELINE config 5 as far as one GPU goes: the per-GPU share of "1 B reads over 8 GPUs" -- 125 M reads of 150 bp
    (7.5 GB of packed reads) -- generated at the read offset of the LAST rank (first_read = 875 000 000, what
    shard_range(10**9, 7, 8) gives), so that 64-bit read indices, the counter-based generator far from zero and a batch
    of this size are all exercised:
      * idempotence (a second pass doubles every count);
      * sharding invariance: five 25 M sub-batches alternating over two slots (two streams, overlapping kernels) give the
        tables of the one 125 M batch;
      * the first 200 k reads of that range are bit-exact against the CPU oracle fed the SAME range from the host generator.
    The exchange between ranks at such offsets: test_gpu_rccl.py::test_two_ranks_at_config5_offsets."""
    from trew_amd.dist import shard_range

    n_total, world, L = 1_000_000_000, 8, 150
    lo, hi = shard_range(n_total, 7, world)
    assert (lo, hi) == (875_000_000, 1_000_000_000)
    n = hi - lo
    stride = 3 * ((L + 31) // 32)
    seed = 20250218
    with T.TrewHip(mode=T.MODE_SHORT, n_slots=2, max_batch_reads=n, max_batch_words=16, table_log2_slots=22) as t:
        d = t.malloc(n * stride * 4 + 64)
        t.synth_short_device(seed, lo, n, L, d)
        t.submit(t.device_uniform_batch(d, n, L))
        t.wait()
        whole = t.collect()
        assert sum(sum(v.values()) for v in whole.values()) > 125_000_000
        t.submit(t.device_uniform_batch(d, n, L), 1)
        t.wait(1)
        twice = t.collect()
        want2 = {name: {k: 2 * c for k, c in whole[name].items()} for name in whole}
        assert twice == want2, _table_diff(twice, want2)
        t.reset_tables()
        q = n // 5
        for i in range(5):
            t.submit(t.device_uniform_batch(d + i * q * stride * 4, q, L), i & 1)
        t.wait(0)
        t.wait(1)
        assert t.collect() == whole
        k = 200_000
        t.reset_tables()
        t.submit(t.device_uniform_batch(d, k, L))
        t.wait()
        got = t.collect()
        t.free(d)
    buf, st, nd = capi.synth_short_ascii(seed, lo, k, L)
    want, _ = O.run_short_mt_timed(O.OracleParams(), buf, st, nd, os.cpu_count() or 8)
    assert got == want
    buf0, _, _ = capi.synth_short_ascii(seed, 0, 64, L)
    assert bytes(buf[:64 * (L + 1)]) != bytes(buf0[:64 * (L + 1)])  # the offset really selects other reads


def test_repeated_passes_are_identical():
    """The prefilter's blocks pull chunks of reads from device counters and the exact kernel's waves pull survivors from a
    sharded queue, so block / wave scheduling differs from launch to launch -- the tables and the survivors must not.  The same
    12 M reads 1200 times, slots alternating: every pass's worklist is compared with the first as a multiset, the tables every
    40th pass.  (A barrier missing at the end of the chunk queue once showed up as a duplicated chunk in about one pass of twelve
    at 125 M reads; a list length read outside its barrier window as a block-full of set-aside reads judged twice and another
    lost, about one pass in three hundred -- tools/flag_diff.py, profiles/r04/README.md.)"""
    n, L, seed = 12_000_000, 150, 20250218
    stride = 3 * ((L + 31) // 32)
    with T.TrewHip(mode=T.MODE_SHORT, n_slots=2, max_batch_reads=n, max_batch_words=16, table_log2_slots=20) as t:
        d = t.malloc(n * stride * 4 + 64)
        t.synth_short_device(seed, 5_000_000_000, n, L, d)  # read indices beyond 2^32 as well
        ref_wl, ref_tab = None, None
        for rep in range(1200):
            t.reset_tables()
            t.submit(t.device_uniform_batch(d, n, L), rep & 1)
            t.wait(rep & 1)
            wl = np.sort(np.asarray(t.debug_worklist(rep & 1), dtype=np.int64))
            if ref_wl is None:
                ref_wl, ref_tab = wl, t.collect()
                assert len(wl) > 150_000 and len(np.unique(wl)) == len(wl)
                assert int(t.last_timing(rep & 1)[2]) == len(wl)
                continue
            assert len(wl) == len(ref_wl) and np.array_equal(wl, ref_wl), (rep, len(wl), len(ref_wl), np.setxor1d(wl, ref_wl)[:8])
            if rep % 40 == 0:
                got = t.collect()
                assert got == ref_tab, (rep, _table_diff(got, ref_tab))
        t.free(d)


@pytest.mark.parametrize("mode", ["short", "pair"])
def test_overlapping_batches_share_the_chip_and_agree(mode):
    """Batches queued on two slots without a wait in between: an exact kernel that finds the other slot busy takes half of
    its wave slots so that the other batch's prefilter is resident beside it (launch_exact, `share`), and the two kernels pull
    from their queues at speeds that differ from pass to pass.  Every pass must give the tables of the same parts run one at
    a time on one slot (tools/stress_overlap.py is the long version: 60 passes of 8 M reads, 30 of 4 M pairs)."""
    L, seed, parts = 150, 20250218, 8
    pair = mode == "pair"
    rp = 2 if pair else 1
    per = 250_000 if pair else 500_000
    stride = 3 * ((L + 31) // 32)
    with T.TrewHip(mode=T.MODE_PAIR if pair else T.MODE_SHORT, n_slots=2, max_batch_reads=rp * per, max_batch_words=16, table_log2_slots=20) as t:
        bufs = []
        for p in range(parts):
            d = t.malloc(rp * per * stride * 4 + 64)
            (t.synth_pair_device if pair else t.synth_short_device)(seed, p * per, per, L, d)
            bufs.append(d)
        for d in bufs:
            t.submit(t.device_uniform_batch(d, rp * per, L), 0)
            t.wait(0)
        ref = t.collect()
        assert sum(len(v) for v in ref.values()) > 1000
        for rep in range(8):
            t.reset_tables()
            for i, d in enumerate(bufs):
                t.submit(t.device_uniform_batch(d, rp * per, L), (i + rep) & 1)
            t.wait(0)
            t.wait(1)
            got = t.collect()
            assert got == ref, (rep, _table_diff(got, ref))
        for d in bufs:
            t.free(d)


def test_full_size_config3_pairs():
    """BASELINE config 3 at full size (50 M pairs of 2 x 150 bp, device-generated, 6 GB of packed reads):
    sharding invariance of the tables, and bit-exactness against the CPU oracle on a 30 k-pair prefix
    (buffer_task_pair, kmer.cpp:268-745)."""
    L = 150
    stride = 3 * ((L + 31) // 32)
    seed = 20250218
    npairs = 50_000_000
    with T.TrewHip(mode=T.MODE_PAIR, max_batch_reads=2 * npairs, max_batch_words=16, table_log2_slots=20) as t:
        d = t.malloc(2 * npairs * stride * 4 + 64)
        t.synth_pair_device(seed, 0, npairs, L, d)
        t.submit(t.device_uniform_batch(d, 2 * npairs, L))
        t.wait()
        whole = t.collect()
        assert sum(sum(v.values()) for v in whole.values()) > 50_000_000
        t.reset_tables()
        q = npairs // 5
        for i in range(5):  # other batch borders, other worklists, two streams
            t.submit(t.device_uniform_batch(d + 2 * i * q * stride * 4, 2 * q, L), i % 2)
        t.wait(0)
        t.wait(1)
        assert t.collect() == whole
        m = 30_000  # pairs checked against the oracle
        t.reset_tables()
        t.submit(t.device_uniform_batch(d, 2 * m, L))
        t.wait()
        got = t.collect()
        t.free(d)
    b1, b2, st, nd = capi.synth_pair_ascii(seed, 0, m, L)
    assert got == O.run_pair(O.OracleParams(), [b1[s:e + 1] for s, e in zip(st, nd)], [b2[s:e + 1] for s, e in zip(st, nd)])


def test_full_size_config4_long_reads():
    """BASELINE config 4 at full size (1 M ONT-like reads, N50 ~ 20 kb, 15.6 Gbases, 5.9 GB of packed reads):
    idempotence and sharding invariance, and bit-exactness against the CPU oracle on the first 6 k reads of
    this very workload (buffer_task_long, kmer.cpp:747-985)."""
    seed = 20250218
    nlong = 1_000_000
    with T.TrewHip(mode=T.MODE_LONG, slice_length=150, max_batch_reads=nlong, max_batch_words=16, table_log2_slots=20) as t:
        b, ptrs, bases = t.synth_long_device(seed, 0, nlong)
        assert bases > 15_000_000_000
        t.submit(b)
        t.wait()
        whole = t.collect()
        assert sum(sum(v.values()) for v in whole.values()) > 5_000_000
        t.submit(b)
        t.wait()
        assert t.collect() == {name: {k: 2 * c for k, c in whole[name].items()} for name in whole}
        # oracle leg: a prefix of the same resident batch
        m = 6_000
        t.reset_tables()
        t.submit(capi.Batch(b.words, b.n_words, b.offsets, b.lengths, 0, 0, m, 1, b.max_length))
        t.wait()
        got = t.collect()
        for p in ptrs:
            t.free(p)
        buf, st, nd = capi.synth_long_ascii(seed, 0, m)
        want = O.run_long(O.OracleParams(), [buf[s:e + 1] for s, e in zip(st, nd)])
        assert got == want
        assert sum(len(v) for v in want.values()) > 100 and len(want["backward_high"]) > 0 and len(want["forward_high"]) > 0
        # four quarters generated separately (first_read offsets the counter-based generator)
        t.reset_tables()
        for i in range(4):
            b2, ptrs2, _ = t.synth_long_device(seed, i * (nlong // 4), nlong // 4)
            t.submit(b2)
            t.wait()
            for p in ptrs2:
                t.free(p)
        assert t.collect() == whole


def _fixed_len_reads(rnd, count, n):
    """Like _fuzz_reads, every read exactly n bases long."""
    from helpers import mutate, periodic

    out = []
    for _ in range(count):
        kind = rnd.random()
        if kind < 0.35:
            s = "".join(rnd.choice("ACGT") for _ in range(n))
            if rnd.random() < 0.2:
                s = mutate(s, rnd, p_sub=0, p_n=rnd.choice([0.005, 0.02, 0.2]))
        elif kind < 0.42:
            s = "".join(rnd.choice("AT") for _ in range(n))
        else:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 70)))
            s = periodic(unit, n, rnd.randint(0, 11))
            s = mutate(s, rnd, p_sub=rnd.choice([0, 0.01, 0.05, 0.2]), p_n=rnd.choice([0, 0, 0, 0.01, 0.1]))
            if rnd.random() < 0.3:
                cut = rnd.randint(0, n)
                u2 = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(2, 40)))
                s = s[:cut] + periodic(u2, n - cut)
        s = s[:n]
        s += "".join(rnd.choice("ACGT") for _ in range(n - len(s)))
        out.append(s.encode())
    return out


@pytest.mark.parametrize("n", [150, 151, 149, 101, 100, 131, 189])
def test_joint_halves_loop_is_sound(n):
    """The prefilter's joint k loop (both halves of a read in one loop, filter_halves_uni; odd lengths judge the longer right half
    by its first L bases against a joint threshold row) is switched off whenever per-k masks are asked for, so
    test_uniform_fast_path_is_sound never sees it.  Here its verdict is observed where it lands: the worklist.  Every read for
    which the oracle records anything -- i.e. some (segment, k) passes k_mer_check -- must be among the flagged units, with the
    joint loop (default) and without it (TREW_FLAG_DEBUG_NO_JOINT); the tables of both runs equal the oracle's."""
    import random

    rnd = random.Random(7700 + n)
    reads = _fixed_len_reads(rnd, 3000, n)
    p = O.OracleParams()
    passing = {i for i, r in enumerate(reads) if any(len(tb) for tb in O.run_short(p, [r]).values())}
    assert len(passing) > 300
    want = O.run_short(p, reads)
    words, offs, lens = capi.pack_reads(reads)
    stride = 3 * ((n + 31) // 32)
    flagged = {}
    for flags in (0, T.FLAG_DEBUG_NO_JOINT):
        with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 20, flags=flags) as t:
            b = capi.Batch(words.ctypes.data, len(words), None, None, n, stride, len(reads), 0, 0)  # uniform batch: the fast path
            t.submit(b, 0)
            t.wait(0)
            wl = t.debug_worklist(0)
            flagged[flags] = set(int(x) for x in wl)
            assert len(flagged[flags]) == len(wl)  # no unit twice
            missing = passing - flagged[flags]
            assert not missing, ("a read with a passing (segment, k) was dropped by the prefilter", flags, sorted(missing)[:5])
            assert t.collect() == want
    # the two loops may flag slightly different false positives (joint threshold of odd lengths), never fewer true ones
    assert len(flagged[0] ^ flagged[T.FLAG_DEBUG_NO_JOINT]) < 0.05 * len(reads)


@pytest.mark.parametrize("n", [150, 151, 126, 190, 142, 66, 96])
def test_uniform_fast_path_is_sound(n):
    """Candidate masks of the prefilter's uniform-geometry fast path, per segment, against the oracle's class counts: every k
    whose MAX/COUNT reaches LOW must be a candidate.  The lengths pick every k-range of the 3-word kernel (halves of 75:
    first-64-windows subset bound for k < 12, one 64-bit container shift for 12 <= k <= 32; halves of 95: three mask words as
    well; halves of 33/48: container and one-word ranges) and the 5-word kernel (n = 190 has halves of 95 -> 3 words, n = 126
    a whole-read segment of 126 bases -> 5 words)."""
    import random

    rnd = random.Random(900 + n)
    reads = _fixed_len_reads(rnd, 1200, n)
    p = O.OracleParams()
    words, offs, lens = capi.pack_reads(reads)
    stride = 3 * ((n + 31) // 32)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 20) as t:
        b = capi.Batch(words.ctypes.data, len(words), None, None, n, stride, len(reads), 0, 0)
        cand = t.filter_masks(b, 3)
    segs = [(0, n // 2, 5, min(n // 4, 32)), (n - (n + 1) // 2, n, 5, min(n // 4, 32))]
    if n < 4 * 32:
        segs.append((0, n, max(n // 4 + 1, 5), min(n // 2, 32)))
    n_pass = 0
    for i, r in enumerate(reads):
        for slot, (a, e, kmin, kmax) in enumerate(segs):
            if kmin > kmax:
                continue
            for k, (cnt, mx, _) in O.segment_stats(p, r[a:e], kmin, kmax).items():
                if cnt and mx / cnt >= 0.5:
                    n_pass += 1
                    assert (int(cand[i, slot]) >> (k - 1)) & 1, (i, slot, k, r)
    assert n_pass > 2000


@pytest.mark.parametrize("seed", range(int(os.environ.get("TREW_FUZZ_SEEDS", "40"))))
def test_fuzz_uniform_batches(seed):
    """Equal-length batches submitted WITHOUT offsets/lengths (uniform_length set): the prefilter takes
    its uniform-geometry fast path (scalar COUNT/masks/thresholds, N reads set aside) -- random
    lengths, k ranges, baselines, motifs, N densities; short and pair mode against the oracle, and the
    same words as a ragged batch (general path) for equality of the two paths."""
    import random

    rnd = random.Random(5000 + seed)
    mn = rnd.choice([3, 4, 5, 6, 9, 17])
    mx = max(mn, rnd.choice([mn, mn + 1, 12, 20, 31, 32, 33, 48, 63, 64]))
    low = rnd.choice([0.5, 0.3, 0.51, 2 / 3, 0.75, 1.0])
    high = max(low, rnd.choice([0.8, 0.6, 0.9, 1.0]))
    kw = dict(min_mer=mn, max_mer=mx, low=low, high=high)
    p = O.OracleParams(**kw)
    n = rnd.choice([rnd.randint(2 * mn, 64), rnd.randint(64, 159), 150, 151, rnd.randint(160, 319), rnd.randint(320, 700)])
    count = 600  # even: also used as 300 pairs
    reads = _fixed_len_reads(rnd, count, n)
    words, offs, lens = capi.pack_reads(reads)
    stride = 3 * ((n + 31) // 32)
    assert len(words) == count * stride and int(offs[1]) == stride
    words = np.ascontiguousarray(words, dtype=np.uint32)
    uni = capi.Batch(words.ctypes.data, len(words), None, None, n, stride, count, 0, n)
    for mode, want in ((T.MODE_SHORT, O.run_short(p, reads)), (T.MODE_PAIR, O.run_pair(p, reads[0::2], reads[1::2]))):
        with T.TrewHip(mode=mode, max_batch_reads=count + 8, max_batch_words=len(words) + 64, **kw) as t:
            t.submit(uni)
            t.wait()
            got = t.collect()
            assert got == want, ("uniform", mode, n, kw)
            t.reset_tables()
            t.submit(t.host_batch(words, offs, lens))
            t.wait()
            assert t.collect() == want, ("ragged", mode, n, kw)


def test_results_do_not_depend_on_lds_residue():
    """Dynamic LDS is not cleared between kernels.  With FLAG_DEBUG_POISON_LDS the exact kernel starts
    from garbage-filled LDS; every mode must still match the oracle (TREW_EXTRA_FLAGS=32 runs the
    whole suite that way)."""
    import random

    rnd = random.Random(77)
    p = O.OracleParams()
    reads = _fuzz_reads(rnd, 400, 300) + edge_reads(3)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22, flags=T.FLAG_DEBUG_POISON_LDS) as t:
        t.submit_reads(reads)
        t.wait()
        assert t.collect() == O.run_short(p, reads)
    r1 = _fuzz_reads(rnd, 200, 300)
    r2 = [_revcomp(r) if rnd.random() < 0.6 else x for r, x in zip(r1, _fuzz_reads(rnd, 200, 300))]
    both = [x for pr in zip(r1, r2) for x in pr]
    with T.TrewHip(mode=T.MODE_PAIR, max_batch_reads=len(both) + 8, max_batch_words=1 << 22, flags=T.FLAG_DEBUG_POISON_LDS) as t:
        t.submit_reads(both)
        t.wait()
        assert t.collect() == O.run_pair(p, r1, r2)
    lr = [r for r in _fuzz_reads(rnd, 150, 4000) if len(r) >= 150]
    with T.TrewHip(mode=T.MODE_LONG, slice_length=150, max_batch_reads=len(lr) + 8, max_batch_words=1 << 22, flags=T.FLAG_DEBUG_POISON_LDS) as t:
        t.submit_reads(lr)
        t.wait()
        assert t.collect() == O.run_long(O.OracleParams(slice_len=150), lr)


def test_tracked_pressure_needs_no_device_query():
    """TREW_FLAG_TRACK_PRESSURE (what the `trew` host runs with): the fill counters travel back with every batch, and
    trew_hip_table_pressure reads those copies -- same numbers as the device query of a context without the flag once the
    batches are waited for, over both slots, across a reset and after rows were added from outside; a tiny table reports
    its spilled rows the same way."""
    buf, st, nd = capi.synth_short_ascii(20250218, 3, 20000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    reads += [r for rep in range(4) for r in edge_reads(30 + rep) if len(r) <= 1000]  # many distinct keys: a 4096-slot table spills
    random.Random(5).shuffle(reads)
    third = len(reads) // 3

    def same(a, b):
        # which of two racing rows of a full partition goes to the spill log is a matter of timing: with the tiny table only
        # the row counts of the tables themselves are comparable, and that both contexts report spilled rows
        pa, pb = a.table_pressure(), b.table_pressure()
        assert pa == pb or (pa[1] == 4096 and pa[0] == pb[0] and pa[2] > 0 and pb[2] > 0 and pa[3] == pb[3]), (pa, pb)

    for log2 in (20, 12):
        with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=third + 8, flags=T.FLAG_TRACK_PRESSURE, table_log2_slots=log2) as a, \
                T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=third + 8, table_log2_slots=log2) as b:
            assert a.table_pressure()[0] == 0
            for t in (a, b):
                t.submit_reads(reads[:third], slot=0)
                t.submit_reads(reads[third:2 * third], slot=1)
                t.wait(0)
                t.wait(1)
            same(a, b)
            assert a.table_pressure()[0] > 0 and (log2 == 20 or a.table_pressure()[2] > 0)
            rows = b.collect()
            a.reset_tables()
            b.reset_tables()
            assert a.table_pressure()[0] == 0 and a.table_pressure()[2] == 0
            for t in (a, b):
                t.submit_reads(reads[2 * third:], slot=1)
                t.wait(1)
            same(a, b)
            before = a.table_pressure()[0]
            a.add_rows(capi.tables_to_rows(rows))
            b.add_rows(capi.tables_to_rows(rows))
            same(a, b)
            assert a.table_pressure()[0] >= before
            assert a.collect() == b.collect()


def test_table_reduction_entry_points():
    """trew_hip_collect_device / trew_hip_add_rows_device / trew_hip_merge / trew_hip_table_pressure: two contexts on
    this GPU scan two halves of a read set; merging one into the other must give the tables of the whole set."""
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 30000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    want = O.run_short(O.OracleParams(), reads)
    half = len(reads) // 2
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=half + 8, max_batch_words=1 << 22) as a, \
            T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=half + 8, max_batch_words=1 << 22) as b:
        a.submit_reads(reads[:half])
        b.submit_reads(reads[half:])
        a.wait()
        b.wait()
        used, total, spilled, spill_cap = a.table_pressure()
        assert used == sum(len(v) for v in a.collect().values()) and total == 1 << 20 and spilled == 0 and spill_cap >= 1 << 16
        part_b = b.collect()
        # (1) in-library merge (peer copy + add kernel); the source is left unchanged
        a.merge_from(b)
        assert a.collect() == want
        assert b.collect() == part_b
        # (2) the same through a caller-owned device buffer (what the RCCL exchange does with torch tensors;
        # torch itself is kept out of this process: tests/test_gpu_rccl.py covers that side)
        a.reset_tables()
        a.submit_reads(reads[:half])
        a.wait()
        row_bytes = capi.ROW_DTYPE.itemsize
        small = a.malloc(4 * row_bytes)
        n = b.collect_device(small, 4)
        assert n == sum(len(v) for v in part_b.values()) > 4  # too small: the size comes back, nothing is written past cap
        a.free(small)
        d_rows = a.malloc(n * row_bytes)
        assert b.collect_device(d_rows, n) == n
        got_rows = a.d2h(d_rows, n * row_bytes).view(capi.ROW_DTYPE)
        assert capi.rows_to_tables(got_rows) == part_b
        a.add_rows_device(d_rows, n)
        assert a.collect() == want
        # a row that cannot be a table row is refused loudly, and the add is all or nothing: the valid rows that travel
        # with the bad one are NOT added, and the context stays usable (no sticky error)
        mixed = got_rows.copy()
        mixed["k"][n // 2] = 99
        a._chk(a.lib.trew_hip_memcpy_h2d(a.ctx, d_rows, mixed.ctypes.data, n * row_bytes), "h2d")
        with pytest.raises(T.TrewHipError, match="nothing was added"):
            a.add_rows_device(d_rows, n)
        assert a.collect() == want
        a._chk(a.lib.trew_hip_memcpy_h2d(a.ctx, d_rows, got_rows.ctypes.data, n * row_bytes), "h2d")
        a.add_rows_device(d_rows, n)  # the same buffer without the bad row: accepted
        assert a.collect() == _tables_sum(want, part_b)
        a.free(d_rows)


def test_gathered_exchange_buffer():
    """trew_hip_add_gathered_device: the one-kernel merge behind the single all_gather of the cross-GPU exchange.  Three
    contexts stand in for three ranks; the gather buffer is assembled by hand (what the collective would deliver)."""
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 45000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    want = O.run_short(O.OracleParams(), reads)
    third = len(reads) // 3
    row_bytes = capi.ROW_DTYPE.itemsize
    ctxs = [T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=third + 8, max_batch_words=1 << 22) for _ in range(3)]
    try:
        for i, c in enumerate(ctxs):
            c.submit_reads(reads[i * third:(i + 1) * third])
            c.wait()
        parts = [c.collect() for c in ctxs]
        n_rows = [sum(len(v) for v in p.values()) for p in parts]
        for cap in (max(n_rows) + 5, max(n_rows)):  # with room to spare, and exactly full
            gathered = ctxs[0].malloc(3 * (1 + cap) * row_bytes)
            hdr = np.zeros(1, dtype=capi.ROW_DTYPE)
            for i, c in enumerate(ctxs):
                base = gathered + i * (1 + cap) * row_bytes
                # rows and header written on the device (trew_hip_collect_slice_device): what allreduce_table_device does
                assert c.collect_slice_device(base, cap, None, want_count=True) == n_rows[i]
                back = np.zeros(1, dtype=capi.ROW_DTYPE)
                c._chk(c.lib.trew_hip_memcpy_d2h(c.ctx, back.ctypes.data, base, row_bytes), "d2h")
                assert (int(back["count"][0]), int(back["k"][0]), int(back["table"][0]), int(back["word_lo"][0])) == (n_rows[i], 0, 0, 0)
            for i, c in enumerate(ctxs):
                assert c.add_gathered_device(gathered, 3, i, cap) == max(n_rows)
                assert c.collect() == want  # every "rank" now holds the global sums
                c.reset_tables()
                c.add_rows(parts[i])
            ctxs[0].free(gathered)
        # a slice too small for one rank's rows: every rank sees the header, nothing is added anywhere, the count comes back
        cap = min(n_rows) - 1
        gathered = ctxs[0].malloc(3 * (1 + cap) * row_bytes)
        for i, c in enumerate(ctxs):
            base = gathered + i * (1 + cap) * row_bytes
            assert c.collect_slice_device(base, cap, None, want_count=True) == n_rows[i]  # the header says how many there ARE
        assert ctxs[1].add_gathered_device(gathered, 3, 1, cap) == max(n_rows) > cap
        assert ctxs[1].collect() == parts[1]
        # a row out of range in a peer's slice: refused, nothing added, context still usable
        cap = max(n_rows)
        ctxs[0].free(gathered)
        gathered = ctxs[0].malloc(3 * (1 + cap) * row_bytes)
        for i, c in enumerate(ctxs):
            base = gathered + i * (1 + cap) * row_bytes
            c.collect_device(base + row_bytes, cap)
            hdr["count"] = n_rows[i]
            c._chk(c.lib.trew_hip_memcpy_h2d(c.ctx, base, hdr.ctypes.data, row_bytes), "h2d")
        bad = np.zeros(1, dtype=capi.ROW_DTYPE)
        bad["k"], bad["table"], bad["count"] = 5, 7, 1
        ctxs[0]._chk(ctxs[0].lib.trew_hip_memcpy_h2d(ctxs[0].ctx, gathered + (2 * (1 + cap) + 3) * row_bytes, bad.ctypes.data, row_bytes), "h2d")
        with pytest.raises(T.TrewHipError, match="nothing was added"):
            ctxs[0].add_gathered_device(gathered, 3, 0, cap)
        assert ctxs[0].collect() == parts[0]
        assert ctxs[2].add_gathered_device(gathered, 3, 2, cap) == max(n_rows)  # rank 2 skips its own (damaged) slice by index
        assert ctxs[2].collect() == want
        ctxs[0].free(gathered)
    finally:
        for c in ctxs:
            c.close()


def test_fallback_paths_are_live_and_exact():
    """The kernels' rare fall-back paths, each forced by a crafted input, counted (trew_hip_debug_counters) and checked
    against the oracle:
      * decide()'s speculative skip refused -> the segment is decided again with every k counted.  (AC)n: k = 5 has one run
        per window (bases i and i+5 always differ) and a 51 % class, it is passed over; k = 6 and 8 are accepted at 1.0,
        k = 10 is accepted too -- a multiple of the skipped 5, which the reference closed when it accepted k = 5
        (kmer.cpp:2225-2236): the check fails and the strict pass must give the reference's answer;
      * eval_runs() with more than 64 runs -> classes counted window by window (that same k = 5: 71 runs in a 75-bp half);
      * the wide table's wait for a slot's ready bit given up (TREW_FLAG_DEBUG_WIDE_NO_WAIT makes every wait time out at
        once): lanes that lose a claim race go on probing and leave duplicate slots, which collect merges."""
    from helpers import periodic

    reads = [periodic("AC", 150, i & 1).encode() for i in range(64)] + [periodic("AG", 121, 0).encode()] * 8
    want = O.run_short(O.OracleParams(), reads)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 20) as t:
        t.submit_reads(reads)
        t.wait()
        got = t.collect()
        c = t.debug_counters()
        assert got == want
        assert c["strict_rerun"] >= 64 and c["windows_fallback"] >= 64 and c["wide_spin_timeout"] == 0, c
        assert c["inserted"] == sum(len(v) for v in got.values()) and c["inserted_wide"] == 0
        t.reset_tables()
        assert all(v == 0 for v in t.debug_counters().values())
    # ordinary reads take none of these paths often: the counters are not a hot path
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 20000, 150)
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=20008, max_batch_words=1 << 22) as t:
        t.submit_reads([buf[s:e + 1] for s, e in zip(st, nd)])
        t.wait()
        c = t.debug_counters()
        assert c["windows_fallback"] < 2000 and c["strict_rerun"] < 2000, c  # 20 000 reads, 450 of them repeats
    # wide keys, many waves inserting the same new keys at once
    import random
    rnd = random.Random(77)
    units = ["".join(rnd.choice("ACGT") for _ in range(k)) for k in (33, 40, 47, 52, 64)]
    wide = [periodic(units[i % 5], 600, (i * 7) % 33).encode() for i in range(3000)]
    want = O.run_short(O.OracleParams(max_mer=64), wide)
    assert any(k > 32 for tb in want.values() for (k, _) in tb)
    with T.TrewHip(mode=T.MODE_SHORT, max_mer=64, max_batch_reads=len(wide) + 8, max_batch_words=1 << 22, flags=T.FLAG_DEBUG_WIDE_NO_WAIT) as t:
        t.submit_reads(wide)
        t.wait()
        assert t.collect() == want
        c = t.debug_counters()
        assert c["wide_spin_timeout"] > 0 and c["inserted_wide"] >= sum(1 for tb in want.values() for (k, _) in tb if k > 32), c


def test_wide_keys_shared_by_a_class_and_its_reverse_complement():
    """k_mer_target_128 (kmer.cpp:2019-2142) keys by MIN(w, rot(rc(w))): a read that holds a k >= 33 repeat and its
    reverse complement emits ONE key from two classes of one wave.  The wide table's claim/ready protocol must never
    make a lane wait for a sibling lane (emit_k merges equal keys first): exact, and fast."""
    import random
    import time

    from helpers import periodic

    rnd = random.Random(4242)
    reads = []
    for i in range(1500):
        k = rnd.randint(33, 64)
        unit = "".join(rnd.choice("ACGT") for _ in range(k))
        rc = _revcomp(unit.encode()).decode()
        n = 1000
        cut = rnd.choice([640, 700, 760])
        reads.append((periodic(unit, cut, rnd.randint(0, k - 1)) + periodic(rc, n - cut, rnd.randint(0, k - 1))).encode())
    want = O.run_short(O.OracleParams(max_mer=64), reads)
    both = want["both_low"]
    assert sum(1 for (k, _) in both if k > 32) > 500  # the case is really there: wide strand-canonical keys
    with T.TrewHip(mode=T.MODE_SHORT, max_mer=64, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22) as t:
        t.submit_reads(reads)  # warm-up (module load)
        t.wait()
        t.reset_tables()
        t0 = time.perf_counter()
        for _ in range(3):
            t.submit_reads(reads)
            t.wait()
        dt = time.perf_counter() - t0
        got = t.collect()
    assert got == {n: {key: 3 * c for key, c in want[n].items()} for n in want}
    assert dt < 5.0, "wide-table inserts took %.1f s: a lane is spinning on a sibling" % dt
