"""End-to-end `trew` binary (FASTQ/.gz reader -> pack -> device scan -> CSV + Putative_TRM) against the oracle."""
import gzip
import os
import subprocess

import pytest

import oracle as O
from oracle.output_oracle import add_totals
from trew_amd import capi
from conftest import GOLDEN, read_fastq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TREW = os.path.join(ROOT, "trew_amd", "bin", "trew")


def write_fastq(path, reads, crlf=False):
    nl = b"\r\n" if crlf else b"\n"
    data = b"".join(b"@r%d" % i + nl + r + nl + b"+" + nl + b"I" * len(r) + nl for i, r in enumerate(reads))
    if path.endswith(".gz"):
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def run(*args):
    r = subprocess.run([TREW, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout.splitlines()


def expected(files_tables, min_mer):
    lines, th, tl = [], {}, {}
    for name, tables in files_tables:
        h, lo = O.fold_tables(tables, min_mer)
        lines += O.format_sections(os.path.realpath(name), h, lo)
        add_totals(th, h)
        add_totals(tl, lo)
    return lines + O.putative_trm(th, tl)


@pytest.mark.parametrize("name", ["test.fastq", "test.fastq.gz"])
def test_cli_fixture_short(name):
    path = os.path.join(GOLDEN, name)
    out = run("short", "5", "32", path)
    assert out == [">H:" + os.path.realpath(path), ">L:" + os.path.realpath(path), ">Putative_TRM", "NO_PUTATIVE_TRM,-1"]


@pytest.mark.parametrize("name", ["test_long.fastq", "test_long.fastq.gz"])
def test_cli_fixture_long(name):
    path = os.path.join(GOLDEN, name)
    out = run("long", "5", "32", path)
    assert out == [">H:" + os.path.realpath(path), ">L:" + os.path.realpath(path), ">Putative_TRM", "NO_PUTATIVE_TRM,-1"]


def test_cli_fixture_short_3_64():
    # the recorded reference output of `short 3 64 test.fastq`, through the 128-bit kernels
    path = os.path.join(GOLDEN, "test.fastq")
    reads = read_fastq(path)
    out = run("short", "3", "64", path)
    assert out == expected([(path, O.run_short(O.OracleParams(min_mer=3, max_mer=64), reads))], 3)
    rows = [r for r in out if r[0].isdigit()]
    assert rows[0] == "3,TTA,157,105,0,-" and "3,TGA,4,+" in out and "3,TGG,3,+" in out


def test_cli_fixture_short_3_32():
    # non-empty rows from the bundled fixture (k = 3 motifs), device limit MAX_MER <= 32
    path = os.path.join(GOLDEN, "test.fastq")
    reads = read_fastq(path)
    p = O.OracleParams(min_mer=3, max_mer=32)
    out = run("short", "3", "32", path, "-t", "3")
    assert out == expected([(path, O.run_short(p, reads))], 3)
    assert "3,TTA,157,105,0,-" in out


def test_cli_short_synthetic_plain_gz_multi_file(tmp_path):
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 60000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    a = str(tmp_path / "a.fastq")
    b = str(tmp_path / "b.fastq.gz")
    write_fastq(a, reads[:40000])
    write_fastq(b, reads[40000:])
    p = O.OracleParams()
    want = expected([(a, O.run_short(p, reads[:40000])), (b, O.run_short(p, reads[40000:]))], 5)
    for threads in ("2", "5"):
        assert run("short", "5", "32", a, b, "-t", threads) == want
    assert any(line.startswith("6,TTAGGG,") for line in want)
    assert want[-1] != "NO_PUTATIVE_TRM,-1"


def test_cli_short_crlf_and_lowercase(tmp_path):
    buf, st, nd = capi.synth_short_ascii(5, 0, 3000, 150)
    reads = [buf[s:e + 1].lower() if i % 3 == 0 else buf[s:e + 1] for i, (s, e) in enumerate(zip(st, nd))]
    a = str(tmp_path / "crlf.fastq")
    write_fastq(a, reads, crlf=True)
    # with CRLF the '\r' belongs to the sequence line (it is an invalid base), as in the reference reader
    want = expected([(a, O.run_short(O.OracleParams(), [r + b"\r" for r in reads]))], 5)
    assert run("short", "5", "32", a) == want


@pytest.mark.parametrize("extra", [[], ["--host_pack"], ["--serial_reader"]])
def test_cli_fastq_of_empty_sequence_lines(tmp_path, extra):
    """A FASTQ whose sequence lines are all empty (one length: zero) used to close a 'uniform' text batch without a length and
    die in trew_hip_submit_ascii; the reference and the CPU-pack path print the empty sections.  All readers must."""
    p = str(tmp_path / "empty_lines.fastq")
    write_fastq(p, [b""] * 500)
    out = run("short", "5", "32", p, "-t", "3", *extra)
    assert out == [">H:" + os.path.realpath(p), ">L:" + os.path.realpath(p), ">Putative_TRM", "NO_PUTATIVE_TRM,-1"]
    # and the same among ordinary reads
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 3000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    mixed = reads[:1500] + [b""] * 40 + reads[1500:]
    p2 = str(tmp_path / "mixed.fastq")
    write_fastq(p2, mixed)
    want = expected([(p2, O.run_short(O.OracleParams(), mixed))], 5)
    assert run("short", "5", "32", p2, "-t", "3", *extra) == want


def test_cli_pair(tmp_path):
    b1, b2, st, nd = capi.synth_pair_ascii(20250218, 0, 30000, 150)
    r1 = [b1[s:e + 1] for s, e in zip(st, nd)]
    r2 = [b2[s:e + 1] for s, e in zip(st, nd)]
    f1 = str(tmp_path / "r1.fastq.gz")
    f2 = str(tmp_path / "r2.fastq")
    write_fastq(f1, r1)
    write_fastq(f2, r2)
    want = expected([(f1, O.run_pair(O.OracleParams(), r1, r2))], 5)
    assert run("short", "5", "32", "--paired_end", "--fq1", f1, "--fq2", f2, "-t", "4") == want


def test_cli_pair_compat_g1(tmp_path):
    """--compat_g1: `trew short --paired_end` with the reference's un-cleared temp_result_left (kmer.cpp:467-505),
    one consumer in file order -- the oracle with compat_g1 on the same pairs, from both readers and with small batches (the
    stale rows cross batch boundaries); without the flag the cleared semantics; with MAX_MER > 32 the flag changes nothing
    (the reference's 128-bit branch clears the map, kmer.cpp:722-723)."""
    from test_gpu_parity import _g1_pairs

    r1, r2 = _g1_pairs(9, 3000)
    f1, f2 = str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")
    write_fastq(f1, r1)
    write_fastq(f2, r2)
    want = expected([(f1, O.run_pair(O.OracleParams(compat_g1=True), r1, r2))], 5)
    cleared = expected([(f1, O.run_pair(O.OracleParams(), r1, r2))], 5)
    assert want != cleared
    base = ["short", "5", "32", "--paired_end", "--fq1", f1, "--fq2", f2]
    for extra in (["-t", "4"], ["-t", "2", "--serial_reader"], ["-t", "8", "--batch_mib", "1"]):
        assert run(*base, "--compat_g1", *extra) == want, extra
    assert run(*base, "-t", "4") == cleared
    wide = expected([(f1, O.run_pair(O.OracleParams(max_mer=40), r1, r2))], 5)
    assert run("short", "5", "40", "--paired_end", "--fq1", f1, "--fq2", f2, "--compat_g1") == wide


def test_cli_pair_block_parallel(tmp_path):
    """Two plain files: the paired block reader (mates located by read index from per-block newline counts, read ranges
    claimed by the workers) against the oracle and against the reference-shaped serial reader; mates of different lengths,
    more pairs than one work item (65 536), a missing final newline in one file still pairs (the counts differ by one
    newline... which the reference reports as a mismatch: so does this reader)."""
    import random

    rnd = random.Random(5)
    b1, b2, st, nd = capi.synth_pair_ascii(20250218, 500, 150000, 150)
    r1 = [b1[s:e + 1][: rnd.choice([150, 150, 140, 101])] for s, e in zip(st, nd)]
    r2 = [b2[s:e + 1][: rnd.choice([150, 150, 150, 128])] for s, e in zip(st, nd)]
    f1, f2 = str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")
    write_fastq(f1, r1)
    write_fastq(f2, r2)
    want = expected([(f1, O.run_pair(O.OracleParams(), r1, r2))], 5)
    for extra in (["-t", "4"], ["-t", "9"], ["-t", "3", "--serial_reader"]):
        assert run("short", "5", "32", "--paired_end", "--fq1", f1, "--fq2", f2, *extra) == want
    r = subprocess.run([TREW, "short", "5", "32", "--paired_end", "--fq1", f1, "--fq2", f2, "-t", "4", "--stats"], capture_output=True, text=True, timeout=300)
    assert "block-parallel paired reader" in r.stderr and "%d reads" % (2 * len(r1)) in r.stderr
    # record counts that differ: the reference's message and exit status (kmer.cpp:1112-1114), from both readers
    f3 = str(tmp_path / "r2_short.fastq")
    write_fastq(f3, r2[:-3])
    for extra in ([], ["--serial_reader"]):
        r = subprocess.run([TREW, "short", "5", "32", "--paired_end", "--fq1", f1, "--fq2", f3, *extra], capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and r.stdout == ""
        assert "Mismatched record counts between files" in r.stderr or (extra and "Paired-end error" in r.stderr)


def test_cli_long(tmp_path):
    from test_gpu_parity import _long_reads

    reads = _long_reads(77, 300)
    a = str(tmp_path / "long.fastq")
    write_fastq(a, reads)
    want = expected([(a, O.run_long(O.OracleParams(), reads))], 5)
    assert run("long", "5", "32", a, "-t", "3") == want
    assert len(want) > 6


@pytest.mark.parametrize("extra", [[], ["--serial_reader"], ["-t", "6"]])
def test_cli_rejects_long_read_in_short_mode(tmp_path, extra):
    """kmer.cpp:1006-1009: message on stderr, exit status 1 -- from a worker thread while other threads hold live
    HIP streams (the process leaves through _exit; it must neither hang nor crash)."""
    buf, st, nd = capi.synth_short_ascii(3, 0, 40000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    reads.insert(30000, b"ACGT" * 300)
    a = str(tmp_path / "x.fastq")
    write_fastq(a, reads)
    r = subprocess.run([TREW, "short", "5", "32", a, *extra], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "This mode is designed for short-read sequencing. Please use 'trew long'." in r.stderr


def test_cli_block_parallel_reader_equals_serial_reader(tmp_path):
    """Plain FASTQ goes through the block-parallel reader (mapped file, 4 MiB blocks claimed by the worker threads);
    --serial_reader forces the reference-shaped single reader.  Same bytes on stdout, several blocks, odd line
    lengths, reads that straddle block borders, with and without a final newline."""
    import random

    rnd = random.Random(12)
    buf, st, nd = capi.synth_short_ascii(20250218, 1000, 50000, 150)
    reads = [buf[s:e + 1][: rnd.choice([150, 150, 150, 149, 101, 36, 0, 7])] for s, e in zip(st, nd)]
    a = str(tmp_path / "a.fastq")
    write_fastq(a, reads)
    assert os.path.getsize(a) > 2 * (1 << 22)
    want = expected([(a, O.run_short(O.OracleParams(), reads))], 5)
    ref = run("short", "5", "32", a, "--serial_reader", "-t", "3")
    assert ref == want
    for t in ("2", "4", "9", "17"):
        assert run("short", "5", "32", a, "-t", t) == want
    assert run("short", "5", "32", a, "-t", "3", "--batch_mib", "1") == want  # several batches per worker
    data = open(a, "rb").read()
    b = str(tmp_path / "b.fastq")
    open(b, "wb").write(data[:-1])  # the last quality line has no newline: nothing changes
    assert run("short", "5", "32", b, "-t", "4") == [x.replace(os.path.realpath(a), os.path.realpath(b)) for x in want]
    r = subprocess.run([TREW, "short", "5", "32", a, "-t", "5", "--stats"], capture_output=True, text=True, timeout=300)
    assert "block-parallel reader" in r.stderr and "%d reads" % len(reads) in r.stderr


def test_cli_long_block_parallel(tmp_path):
    from test_gpu_parity import _long_reads

    reads = _long_reads(78, 3000)  # ~ 2 blocks, reads of up to 12 kb straddle the border; short ones are dropped (kmer.cpp:1184)
    a = str(tmp_path / "long.fastq")
    write_fastq(a, reads)
    assert os.path.getsize(a) > (1 << 22)
    want = expected([(a, O.run_long(O.OracleParams(), reads))], 5)
    assert run("long", "5", "32", a, "-t", "4") == want
    assert run("long", "5", "32", a, "-t", "4", "--serial_reader") == want


def test_cli_tiny_table_is_drained_not_lost(tmp_path):
    """The device count table is fixed-size where the reference's hash maps grow (kmer.h:79).  With a 4096-slot table
    these reads cannot fit: the host must empty the table into memory mid-file (trew_hip_table_pressure ->
    collect -> reset) and still print the oracle's CSV."""
    from helpers import edge_reads, mixed_segments

    reads = []
    for rep in range(12):  # several 4 MiB blocks, each with far more keys than the table has slots
        reads += [r for r in edge_reads(11 + rep) if len(r) <= 1000] + mixed_segments(5 + rep, 1500, [150, 200, 300])
    a = str(tmp_path / "keys.fastq")
    write_fastq(a, reads)
    tables = O.run_short(O.OracleParams(), reads)
    assert sum(len(v) for v in tables.values()) > 3 * 4096
    want = expected([(a, tables)], 5)
    for extra in (["-t", "2", "--batch_mib", "1"], ["-t", "2", "--serial_reader"]):  # one worker: the drain points are deterministic
        r = subprocess.run([TREW, "short", "5", "32", a, "--table_log2_slots", "12", "--stats", *extra], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert r.stdout.splitlines() == want
        drains = int(r.stderr.split(" table drain(s)")[0].split()[-1])
        assert drains >= 1, r.stderr
    assert run("short", "5", "32", a) == want  # default table: same output, no drain needed
    # several contexts: each is kept at or below half full during the file, but their UNION does not fit device 0 --
    # the final reduction (trew_hip_merge into device 0) must drain device 0 between merges instead of overflowing it
    r = subprocess.run([TREW, "short", "5", "32", a, "--table_log2_slots", "12", "--devices", "0,0,0", "-t", "4", "--batch_mib", "1", "--stats"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == want


def test_cli_several_contexts_reduce_on_the_device(tmp_path):
    """--devices LIST: one context per entry, worker threads spread over them, tables reduced with trew_hip_merge
    (peer copy + add kernel) before they are collected.  One GPU here, so the list names it twice."""
    buf, st, nd = capi.synth_short_ascii(20250218, 7, 50000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    a = str(tmp_path / "a.fastq")
    write_fastq(a, reads)
    want = expected([(a, O.run_short(O.OracleParams(), reads))], 5)
    assert run("short", "5", "32", a, "--devices", "0,0", "-t", "5") == want
    assert run("short", "5", "32", a, "--devices", "0,0,0", "-t", "4", "--serial_reader") == want


def test_cli_bgzf_input(tmp_path):
    """Block-gzip (bgzip) input goes through the parallel BGZF reader: same output as the plain file,
    single and paired, with members cut at arbitrary places (lines and records span members)."""
    from test_bgzf_cpu import write_bgzf

    b1, b2, st, nd = capi.synth_pair_ascii(20250218, 0, 40000, 150)
    r1 = [b1[s:e + 1] for s, e in zip(st, nd)]
    r2 = [b2[s:e + 1] for s, e in zip(st, nd)]
    plain1, plain2 = str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")
    write_fastq(plain1, r1)
    write_fastq(plain2, r2)
    z1, z2 = str(tmp_path / "r1.fastq.gz"), str(tmp_path / "r2.fastq.gz")
    write_bgzf(z1, open(plain1, "rb").read())
    write_bgzf(z2, open(plain2, "rb").read(), block=12345)
    want_single = expected([(z1, O.run_short(O.OracleParams(), r1))], 5)
    assert run("short", "5", "32", z1, "-t", "4") == want_single
    want_pair = expected([(z1, O.run_pair(O.OracleParams(), r1, r2))], 5)
    assert run("short", "5", "32", "--paired_end", "--fq1", z1, "--fq2", z2, "-t", "4") == want_pair


@pytest.mark.parametrize("block", [0xFF00, 4000, 777])
def test_cli_bgzf_block_parallel_reader(tmp_path, block):
    """BGZF input, single file: members are indexed from their headers, worker threads inflate groups of them into place and
    locate the sequence lines there (run_blocks_bgzf) -- the same CSV as the plain file through the block reader and as the same
    BGZF file through the serial reader, with members of 64 KiB, 4 000 and 777 bytes (lines and records span members and
    blocks), CRLF line ends, reads of unequal length, and in long mode."""
    from helpers import periodic
    from test_bgzf_cpu import write_bgzf

    import random

    rnd = random.Random(31 + block)
    buf, st, nd = capi.synth_short_ascii(20250218, 100, 60000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    reads += [periodic("TTAGGG", rnd.randint(20, 900), rnd.randint(0, 5)).encode() for _ in range(400)] + [b"", b"ACGT"]
    rnd.shuffle(reads)
    plain, z = str(tmp_path / "r.fastq"), str(tmp_path / "r.fastq.gz")
    write_fastq(plain, reads, crlf=(block == 4000))
    write_bgzf(z, open(plain, "rb").read(), block=block)
    if block == 4000:  # CRLF: the carriage return is part of the sequence line, as in the reference
        reads = [r + b"\r" for r in reads]
    want = expected([(z, O.run_short(O.OracleParams(), reads))], 5)
    r = subprocess.run([TREW, "short", "5", "32", z, "-t", "5", "--stats"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "block-parallel BGZF reader" in r.stderr
    assert r.stdout.splitlines() == want
    assert run("short", "5", "32", z, "-t", "5", "--serial_reader") == want
    assert run("short", "5", "32", z, "-t", "3", "--host_pack", "--batch_mib", "1") == want


def test_cli_bgzf_block_parallel_long_reads_and_fallbacks(tmp_path):
    """Long reads (lines of tens of kilobases across many members and across blocks) through the BGZF block reader; a file that
    is BGZF followed by a plain gzip member, and a BGZF file with a damaged member, take the serial reader's way (same
    output / same error)."""
    import gzip as gz

    from test_bgzf_cpu import write_bgzf
    from test_gpu_parity import _long_reads

    lr = [r for r in _long_reads(11, 150) if len(r) >= 150]
    plain, z = str(tmp_path / "l.fastq"), str(tmp_path / "l.fastq.gz")
    write_fastq(plain, lr)
    write_bgzf(z, open(plain, "rb").read(), block=30000)
    want = expected([(z, O.run_long(O.OracleParams(slice_len=150), lr))], 5)
    r = subprocess.run([TREW, "long", "5", "32", z, "-t", "4", "--stats"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "block-parallel BGZF reader" in r.stderr, r.stderr
    assert r.stdout.splitlines() == want
    # BGZF + a plain gzip member: not BGZF from end to end -> serial reader, which reads both parts (as gzread does)
    buf, st, nd = capi.synth_short_ascii(20250218, 0, 6000, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    pa, pb, mixed = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq"), str(tmp_path / "mixed.fastq.gz")
    write_fastq(pa, reads[:3000])
    write_fastq(pb, reads[3000:])
    za = str(tmp_path / "a.gz")
    write_bgzf(za, open(pa, "rb").read(), block=5000)
    data = open(za, "rb").read()
    data = data[:-28] if data.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")) else data  # drop the empty EOF member if there is one
    open(mixed, "wb").write(data + gz.compress(open(pb, "rb").read()))
    want = expected([(mixed, O.run_short(O.OracleParams(), reads))], 5)
    r = subprocess.run([TREW, "short", "5", "32", mixed, "-t", "4", "--stats"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "serial reader" in r.stderr, r.stderr
    assert r.stdout.splitlines() == want
    # a damaged member (a byte of its deflate stream flipped): an error either way, never a silent difference
    bad = bytearray(open(za, "rb").read())
    bad[len(bad) // 2] ^= 0x5A
    pbad = str(tmp_path / "bad.fastq.gz")
    open(pbad, "wb").write(bytes(bad))
    r = subprocess.run([TREW, "short", "5", "32", pbad, "-t", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "Error" in r.stderr


def _rank_scan(rank, world, port, n, q):
    import torch.distributed as dist

    import trew_amd as T
    from trew_amd.dist import allreduce_rows_into_table, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n, rank, world)
    buf, st, nd = capi.synth_short_ascii(20250218, lo, hi - lo, 150)
    reads = [buf[s:e + 1] for s, e in zip(st, nd)]
    with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=hi - lo + 8, max_batch_words=1 << 22) as t:
        t.submit_reads(reads)
        t.wait()
        merged = allreduce_rows_into_table(t, t.collect_rows())
    q.put((rank, capi.rows_to_tables(merged)))
    dist.destroy_process_group()


def test_two_ranks_reduce_through_the_device_table():
    """The reduction bench.py uses for N > 1 (all_gather of rows, add_rows into the device table, collect),
    with two processes sharing this GPU over gloo: every rank must end with the tables of all reads."""
    import torch.multiprocessing as mp

    n, world = 40000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_rank_scan, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    buf, st, nd = capi.synth_short_ascii(20250218, 0, n, 150)
    want = O.run_short(O.OracleParams(), [buf[s:e + 1] for s, e in zip(st, nd)])
    for rank, got in res:
        assert got == want, rank
