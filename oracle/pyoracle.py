"""ctypes wrapper around oracle/libtrew_oracle.so -- TEST INFRASTRUCTURE ONLY.

The C side (trew_oracle.c) is the CPU restatement of the reference algorithm
(Chemical118/TREW src/kmer.cpp); this file only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtrew_oracle.so")
_lib = None

TABLE_NAMES = ("forward_high", "forward_low", "backward_high", "backward_low", "both_high", "both_low")


class _Params(C.Structure):
    _fields_ = [
        ("min_mer", C.c_int),
        ("max_mer", C.c_int),
        ("low", C.c_double),
        ("high", C.c_double),
        ("slice_len", C.c_int),
        ("use_break", C.c_int),
        ("compat_g1", C.c_int),
    ]


class _Row(C.Structure):
    _fields_ = [
        ("k", C.c_int32),
        ("pad", C.c_int32),
        ("word_lo", C.c_uint64),
        ("word_hi", C.c_uint64),
        ("count", C.c_uint64),
    ]


@dataclass
class OracleParams:
    min_mer: int = 5
    max_mer: int = 32
    low: float = 0.5
    high: float = 0.8
    slice_len: int = 150
    use_break: bool = True
    compat_g1: bool = False  # pairs: the 64-bit branch's un-cleared temp_result_left, one context, file order

    def c(self) -> _Params:
        return _Params(self.min_mer, self.max_mer, self.low, self.high, self.slice_len, 1 if self.use_break else 0, 1 if self.compat_g1 else 0)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no GPU, no reference sources involved)."""
    srcs = [os.path.join(_HERE, f) for f in ("trew_oracle.c", "trew_oracle_core.inc", "trew_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s)
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = C.CDLL(_LIB_PATH)
    u64p = C.POINTER(C.c_uint64)
    i64p = C.POINTER(C.c_int64)
    lib.trew_oracle_code.restype = C.c_int
    lib.trew_oracle_code.argtypes = [C.c_ubyte]
    lib.trew_oracle_rot_seq.argtypes = [C.c_uint64, C.c_uint64, C.c_int, u64p, u64p]
    lib.trew_oracle_revcomp.argtypes = [C.c_uint64, C.c_uint64, C.c_int, u64p, u64p]
    lib.trew_oracle_repeat_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    lib.trew_oracle_repeat_check.restype = C.c_int
    lib.trew_oracle_check_ans_seq.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    lib.trew_oracle_check_ans_seq.restype = C.c_int
    lib.trew_oracle_segment_check.argtypes = [
        C.POINTER(_Params), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
        C.POINTER(C.c_int), C.POINTER(C.c_int), u64p, u64p,
        C.POINTER(_Row), C.POINTER(C.c_int), C.POINTER(_Row), C.POINTER(C.c_int), C.c_int,
    ]
    lib.trew_oracle_segment_stats.argtypes = [
        C.POINTER(_Params), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u64p, u64p,
    ]
    lib.trew_oracle_new.restype = C.c_void_p
    lib.trew_oracle_new.argtypes = [C.POINTER(_Params)]
    lib.trew_oracle_free.argtypes = [C.c_void_p]
    lib.trew_oracle_add_short.argtypes = [C.c_void_p, C.c_char_p, i64p, i64p, C.c_int64]
    lib.trew_oracle_add_long.argtypes = [C.c_void_p, C.c_char_p, i64p, i64p, C.c_int64]
    lib.trew_oracle_add_pair.argtypes = [C.c_void_p, C.c_char_p, i64p, i64p, C.c_char_p, i64p, i64p, C.c_int64]
    lib.trew_oracle_table_size.restype = C.c_int64
    lib.trew_oracle_table_size.argtypes = [C.c_void_p, C.c_int]
    lib.trew_oracle_table_rows.restype = C.c_int64
    lib.trew_oracle_table_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Row), C.c_int64]
    lib.trew_oracle_run_short_mt.restype = C.c_void_p
    lib.trew_oracle_run_short_mt.argtypes = [C.POINTER(_Params), C.c_char_p, i64p, i64p, C.c_int64, C.c_int]
    _lib = lib
    return lib


_CODE = {"T": 0, "G": 1, "C": 2, "A": 3}
_LETTER = "TGCA"


def four_to_int(s: str) -> int:
    """test.cpp:63-81 four_to_int: first base most significant."""
    v = 0
    for ch in s:
        v = v * 4 + _CODE[ch.upper()]
    return v


def int_to_four(word: int, k: int) -> str:
    """int_to_four, kmer.cpp:1886-1892."""
    return "".join(_LETTER[(word >> (2 * (k - 1 - i))) & 3] for i in range(k))


def code(ch: str) -> int:
    return _load().trew_oracle_code(ord(ch))


def _split(w: int):
    return w & 0xFFFFFFFFFFFFFFFF, (w >> 64) & 0xFFFFFFFFFFFFFFFF


def rot_seq(word: int, k: int) -> int:
    lo, hi = C.c_uint64(), C.c_uint64()
    a, b = _split(word)
    _load().trew_oracle_rot_seq(a, b, k, C.byref(lo), C.byref(hi))
    return (hi.value << 64) | lo.value


def revcomp(word: int, k: int) -> int:
    lo, hi = C.c_uint64(), C.c_uint64()
    a, b = _split(word)
    _load().trew_oracle_revcomp(a, b, k, C.byref(lo), C.byref(hi))
    return (hi.value << 64) | lo.value


def repeat_check(word: int, k: int) -> int:
    a, b = _split(word)
    return _load().trew_oracle_repeat_check(a, b, k)


def check_ans_seq(word: int, k: int, min_mer: int) -> bool:
    a, b = _split(word)
    return bool(_load().trew_oracle_check_ans_seq(a, b, k, min_mer))


def _rows_to_dict(rows, n):
    return {(rows[i].k, (rows[i].word_hi << 64) | rows[i].word_lo): int(rows[i].count) for i in range(n)}


def segment_check(p: OracleParams, seq: bytes, min_mer: int | None = None, max_mer: int | None = None):
    """k_mer_check on the whole of `seq`.  Returns dict(k_high, k_low, seq_high,
    seq_low, hist_high, hist_low) with histograms as {(k, word): count}."""
    lib = _load()
    if isinstance(seq, str):
        seq = seq.encode()
    min_mer = p.min_mer if min_mer is None else min_mer
    max_mer = p.max_mer if max_mer is None else max_mer
    cap = max(16, len(seq) + 1)
    kh, kl = C.c_int(), C.c_int()
    sh = (C.c_uint64 * 2)()
    sl = (C.c_uint64 * 2)()
    hh = (_Row * cap)()
    hl = (_Row * cap)()
    nh, nl = C.c_int(), C.c_int()
    pc = p.c()
    lib.trew_oracle_segment_check(C.byref(pc), seq, 0, len(seq) - 1, min_mer, max_mer, C.byref(kh), C.byref(kl),
                                  sh, sl, hh, C.byref(nh), hl, C.byref(nl), cap)
    return dict(
        k_high=kh.value, k_low=kl.value,
        seq_high=(sh[1] << 64) | sh[0], seq_low=(sl[1] << 64) | sl[0],
        hist_high=_rows_to_dict(hh, nh.value), hist_low=_rows_to_dict(hl, nl.value),
    )


def segment_stats(p: OracleParams, seq: bytes, min_mer: int, max_mer: int):
    """Per-k (COUNT, MAX, MAX_SEQ) of one segment, no early break."""
    lib = _load()
    if isinstance(seq, str):
        seq = seq.encode()
    nk = max_mer - min_mer + 1
    cnt = (C.c_uint32 * nk)()
    mx = (C.c_uint32 * nk)()
    lo = (C.c_uint64 * nk)()
    hi = (C.c_uint64 * nk)()
    pc = p.c()
    lib.trew_oracle_segment_stats(C.byref(pc), seq, 0, len(seq) - 1, min_mer, max_mer, cnt, mx, lo, hi)
    return {min_mer + i: (cnt[i], mx[i], (hi[i] << 64) | lo[i]) for i in range(nk)}


def _concat(reads):
    """Lay reads out back to back with a '\\n' guard between them, as in a FASTQ chunk."""
    st = np.zeros(len(reads), dtype=np.int64)
    nd = np.zeros(len(reads), dtype=np.int64)
    parts = []
    pos = 0
    for i, r in enumerate(reads):
        if isinstance(r, str):
            r = r.encode()
        st[i] = pos
        nd[i] = pos + len(r) - 1
        parts.append(r)
        parts.append(b"\n")
        pos += len(r) + 1
    return b"".join(parts), st, nd


def _tables(lib, ctx):
    out = {}
    for t, name in enumerate(TABLE_NAMES):
        n = lib.trew_oracle_table_size(ctx, t)
        rows = (_Row * max(1, n))()
        n = lib.trew_oracle_table_rows(ctx, t, rows, n)
        out[name] = _rows_to_dict(rows, n)
    return out


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def run_short(p: OracleParams, reads):
    """buffer_task over a list of reads -> the six tables {name: {(k, word): count}}."""
    lib = _load()
    buf, st, nd = _concat(reads)
    pc = p.c()
    ctx = lib.trew_oracle_new(C.byref(pc))
    lib.trew_oracle_add_short(ctx, buf, _i64p(st), _i64p(nd), len(reads))
    out = _tables(lib, ctx)
    lib.trew_oracle_free(ctx)
    return out


def run_long(p: OracleParams, reads):
    lib = _load()
    reads = [r for r in reads if len(r) >= p.slice_len]  # read_fastq_long_thread drops them, kmer.cpp:1184
    buf, st, nd = _concat(reads)
    pc = p.c()
    ctx = lib.trew_oracle_new(C.byref(pc))
    lib.trew_oracle_add_long(ctx, buf, _i64p(st), _i64p(nd), len(reads))
    out = _tables(lib, ctx)
    lib.trew_oracle_free(ctx)
    return out


def run_pair(p: OracleParams, reads1, reads2):
    lib = _load()
    n = min(len(reads1), len(reads2))
    b1, s1, e1 = _concat(reads1[:n])
    b2, s2, e2 = _concat(reads2[:n])
    pc = p.c()
    ctx = lib.trew_oracle_new(C.byref(pc))
    lib.trew_oracle_add_pair(ctx, b1, _i64p(s1), _i64p(e1), b2, _i64p(s2), _i64p(e2), n)
    out = _tables(lib, ctx)
    lib.trew_oracle_free(ctx)
    return out


def run_short_mt_timed(p: OracleParams, buf: bytes, st: np.ndarray, nd: np.ndarray, nthreads: int):
    """Timed multi-threaded short-mode run over pre-laid-out reads.  Returns
    (tables, seconds).  Used for bench.py's cpu_baseline ("port")."""
    lib = _load()
    pc = p.c()
    st = np.ascontiguousarray(st, dtype=np.int64)
    nd = np.ascontiguousarray(nd, dtype=np.int64)
    t0 = time.perf_counter()
    ctx = lib.trew_oracle_run_short_mt(C.byref(pc), buf, _i64p(st), _i64p(nd), len(st), nthreads)
    dt = time.perf_counter() - t0
    out = _tables(lib, ctx)
    lib.trew_oracle_free(ctx)
    return out, dt
