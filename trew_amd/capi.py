"""ctypes binding of include/trew_hip.h (libtrew_hip.so).

This is plumbing for tests and bench.py; the product is the HIP library.  There
is no CPU fallback here: if the library is missing or no GPU is present, the
compute entry points raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TREW_HIP_LIB") or os.path.join(_HERE, "lib", "libtrew_hip.so")  # TREW_HIP_LIB: alternative build (tools/phase_profile.py)

MODE_SHORT, MODE_PAIR, MODE_LONG, MODE_SEGMENT = 0, 1, 2, 3
FLAG_NO_FILTER = 1
FLAG_DEBUG_POISON_LDS = 32  # the exact kernel starts from garbage-filled LDS (tests)
FLAG_NO_TIMING = 64  # no HIP events per submit (last_timing unavailable)
FLAG_COMPAT_G1 = 512  # pair mode, one slot: the reference's 64-bit pair branch as written (un-cleared temp_result_left)
FLAG_TRACK_PRESSURE = 256  # the fill counters come back with every batch; table_pressure asks no device
FLAG_DEBUG_NO_JOINT = 2048  # tests: the prefilter's uniform path without the joint k loop of both halves
FLAG_DEBUG_NO_GROUP = 1024  # tests / A-B: every segment decided by a wave of its own (no decide_group)
FLAG_DEBUG_WIDE_NO_WAIT = 128  # tests: the wide table never waits for a slot's ready bit (forces its time-out path)
TABLE_NAMES = ("forward_high", "forward_low", "backward_high", "backward_low", "both_high", "both_low")

# every symbol include/trew_hip.h declares
EXPORTED_SYMBOLS = (
    "trew_hip_init", "trew_hip_destroy", "trew_hip_last_error", "trew_hip_submit", "trew_hip_wait",
    "trew_hip_collect", "trew_hip_reset_tables", "trew_hip_add_rows", "trew_hip_segment_results",
    "trew_hip_filter_masks", "trew_hip_last_timing", "trew_pack_words", "trew_pack_reads",
    "trew_synth_short_ascii", "trew_synth_short_device", "trew_synth_pair_ascii", "trew_synth_pair_device",
    "trew_hip_malloc", "trew_hip_free", "trew_hip_memcpy_h2d", "trew_hip_memcpy_d2h", "trew_hip_abi_version",
    "trew_pack_pairs", "trew_hip_host_alloc", "trew_hip_host_free", "trew_hip_device_count",
    "trew_synth_long_lengths", "trew_synth_long_ascii", "trew_synth_long_device",
    "trew_hip_collect_device", "trew_hip_add_rows_device", "trew_hip_merge", "trew_hip_table_pressure",
    "trew_hip_add_gathered_device", "trew_hip_collect_slice_device", "trew_hip_debug_counters", "trew_hip_debug_worklist", "trew_hip_submit_ascii", "trew_hip_pack_ascii",
)
DEBUG_COUNTERS = ("strict_rerun", "windows_fallback", "wide_spin_timeout", "inserted", "inserted_wide", "group_punt", "group_routed", "group_target")


class Params(C.Structure):
    _fields_ = [
        ("min_mer", C.c_int32), ("max_mer", C.c_int32),
        ("low_baseline", C.c_double), ("high_baseline", C.c_double),
        ("slice_length", C.c_int32), ("mode", C.c_int32), ("device", C.c_int32), ("n_slots", C.c_int32),
        ("max_batch_words", C.c_uint64), ("max_batch_reads", C.c_uint64),
        ("table_log2_slots", C.c_uint32), ("flags", C.c_uint32),
        ("max_batch_ascii_bytes", C.c_uint64),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("words", C.c_void_p), ("n_words", C.c_uint64),
        ("offsets", C.c_void_p), ("lengths", C.c_void_p),
        ("uniform_length", C.c_uint32), ("uniform_stride", C.c_uint32),
        ("n_reads", C.c_uint64), ("on_device", C.c_int32), ("max_length", C.c_int32),
    ]


class AsciiBatch(C.Structure):
    _fields_ = [
        ("bases", C.c_void_p), ("n_bytes", C.c_uint64),
        ("byte_offsets", C.c_void_p), ("lengths", C.c_void_p), ("word_offsets", C.c_void_p),
        ("uniform_length", C.c_uint32), ("reserved", C.c_uint32), ("n_reads", C.c_uint64),
    ]


class Row(C.Structure):
    _fields_ = [("k", C.c_int32), ("table", C.c_int32), ("word_lo", C.c_uint64), ("word_hi", C.c_uint64),
                ("count", C.c_uint64)]


ROW_DTYPE = np.dtype([("k", "<i4"), ("table", "<i4"), ("word_lo", "<u8"), ("word_hi", "<u8"), ("count", "<u8")])
assert ROW_DTYPE.itemsize == C.sizeof(Row)

_lib = None


def rows_to_tables(rows):
    """Structured row array -> {name: {(k, word): count}}."""
    out = {name: {} for name in TABLE_NAMES}
    if len(rows) == 0:
        return out
    for t, k, lo, hi, c in zip(rows["table"].tolist(), rows["k"].tolist(), rows["word_lo"].tolist(),
                               rows["word_hi"].tolist(), rows["count"].tolist()):
        out[TABLE_NAMES[t]][(k, (hi << 64) | lo)] = c
    return out


def tables_to_rows(tables):
    items = [(k, t, w & 0xFFFFFFFFFFFFFFFF, w >> 64, c)
             for t, name in enumerate(TABLE_NAMES) for (k, w), c in tables.get(name, {}).items()]
    return np.array(items, dtype=ROW_DTYPE) if items else np.zeros(0, dtype=ROW_DTYPE)


def load():
    """Load libtrew_hip.so (raises OSError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libtrew_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    lib.trew_hip_abi_version.restype = i32
    lib.trew_hip_init.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    lib.trew_hip_destroy.argtypes = [vp]
    lib.trew_hip_destroy.restype = None
    lib.trew_hip_last_error.argtypes = [vp]
    lib.trew_hip_last_error.restype = C.c_char_p
    lib.trew_hip_submit.argtypes = [vp, C.POINTER(Batch), i32]
    lib.trew_hip_wait.argtypes = [vp, i32]
    lib.trew_hip_submit_ascii.argtypes = [vp, C.POINTER(AsciiBatch), i32]
    lib.trew_hip_pack_ascii.argtypes = [vp, C.POINTER(AsciiBatch), vp, u64, C.POINTER(u64)]
    lib.trew_hip_collect.argtypes = [vp, i32, C.POINTER(Row), u64, C.POINTER(u64)]
    lib.trew_hip_reset_tables.argtypes = [vp]
    lib.trew_hip_add_rows.argtypes = [vp, C.POINTER(Row), u64]
    lib.trew_hip_collect_device.argtypes = [vp, vp, u64, C.POINTER(u64)]
    lib.trew_hip_add_rows_device.argtypes = [vp, vp, u64]
    lib.trew_hip_merge.argtypes = [vp, vp]
    lib.trew_hip_add_gathered_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, u64, vp, C.POINTER(u64)]
    lib.trew_hip_collect_slice_device.argtypes = [vp, vp, u64, vp, C.POINTER(u64)]
    lib.trew_hip_debug_counters.argtypes = [vp, C.POINTER(u64), i32]
    lib.trew_hip_debug_worklist.argtypes = [vp, i32, vp, u64, C.POINTER(u64)]
    lib.trew_hip_table_pressure.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.trew_hip_segment_results.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, u64]
    lib.trew_hip_filter_masks.argtypes = [vp, C.POINTER(Batch), vp, i32]
    lib.trew_hip_last_timing.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(u64)]
    lib.trew_pack_words.argtypes = [u64]
    lib.trew_pack_words.restype = u64
    lib.trew_pack_reads.argtypes = [C.c_char_p, vp, vp, u64, vp, u64, vp, vp]
    lib.trew_pack_reads.restype = u64
    lib.trew_synth_short_ascii.argtypes = [u64, u64, u64, C.c_uint32, vp]
    lib.trew_synth_short_device.argtypes = [vp, u64, u64, u64, C.c_uint32, vp]
    lib.trew_synth_pair_ascii.argtypes = [u64, u64, u64, C.c_uint32, vp, vp]
    lib.trew_synth_pair_device.argtypes = [vp, u64, u64, u64, C.c_uint32, vp]
    lib.trew_synth_long_lengths.argtypes = [u64, u64, u64, vp]
    lib.trew_synth_long_ascii.argtypes = [u64, u64, u64, vp, vp]
    lib.trew_synth_long_device.argtypes = [vp, u64, u64, u64, vp, vp]
    lib.trew_hip_malloc.argtypes = [vp, u64, C.POINTER(vp)]
    lib.trew_hip_free.argtypes = [vp, vp]
    lib.trew_hip_memcpy_h2d.argtypes = [vp, vp, vp, u64]
    lib.trew_hip_memcpy_d2h.argtypes = [vp, vp, vp, u64]
    _lib = lib
    return lib


class TrewHipError(RuntimeError):
    pass


def pack_reads(reads):
    """codes[] (kmer.cpp:14-31) applied on the host: list of byte strings -> (words, offsets, lengths)."""
    lib = load()
    reads = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    n = len(reads)
    st = np.zeros(n, dtype=np.int64)
    nd = np.zeros(n, dtype=np.int64)
    pos = 0
    for i, r in enumerate(reads):
        st[i] = pos
        nd[i] = pos + len(r) - 1
        pos += len(r) + 1
    buf = b"\n".join(reads) + b"\n"
    cap = int(sum(lib.trew_pack_words(len(r)) for r in reads)) + 8
    words = np.zeros(cap, dtype=np.uint32)
    offsets = np.zeros(max(n, 1), dtype=np.uint32)
    lengths = np.zeros(max(n, 1), dtype=np.uint32)
    w = lib.trew_pack_reads(buf, st.ctypes.data, nd.ctypes.data, n, words.ctypes.data, cap, offsets.ctypes.data,
                            lengths.ctypes.data)
    if w == 2 ** 64 - 1:
        raise TrewHipError("trew_pack_reads: buffer too small")
    return words[: int(w)], offsets[:n], lengths[:n]


def synth_short_ascii(seed, first_read, n_reads, read_len):
    """Host side of the synthetic short-read generator: returns (buf, st, nd)."""
    lib = load()
    out = np.zeros(n_reads * (read_len + 1), dtype=np.uint8)
    lib.trew_synth_short_ascii(seed, first_read, n_reads, read_len, out.ctypes.data)
    st = np.arange(n_reads, dtype=np.int64) * (read_len + 1)
    nd = st + read_len - 1
    return out.tobytes(), st, nd


def synth_pair_ascii(seed, first_pair, n_pairs, read_len):
    lib = load()
    o1 = np.zeros(n_pairs * (read_len + 1), dtype=np.uint8)
    o2 = np.zeros(n_pairs * (read_len + 1), dtype=np.uint8)
    lib.trew_synth_pair_ascii(seed, first_pair, n_pairs, read_len, o1.ctypes.data, o2.ctypes.data)
    st = np.arange(n_pairs, dtype=np.int64) * (read_len + 1)
    nd = st + read_len - 1
    return o1.tobytes(), o2.tobytes(), st, nd


def synth_long_lengths(seed, first_read, n_reads):
    lib = load()
    lens = np.zeros(n_reads, dtype=np.uint32)
    lib.trew_synth_long_lengths(seed, first_read, n_reads, lens.ctypes.data)
    return lens


def synth_long_ascii(seed, first_read, n_reads):
    """Host side of the long-read generator: (buf, st, nd) with one '\\n' after each read."""
    lib = load()
    lens = synth_long_lengths(seed, first_read, n_reads).astype(np.int64)
    st = np.zeros(n_reads, dtype=np.int64)
    st[1:] = np.cumsum(lens[:-1] + 1)
    out = np.zeros(int(st[-1] + lens[-1] + 1) if n_reads else 0, dtype=np.uint8)
    st_u64 = np.ascontiguousarray(st, dtype=np.uint64)  # keep the array alive across the call
    lib.trew_synth_long_ascii(seed, first_read, n_reads, st_u64.ctypes.data, out.ctypes.data)
    return out.tobytes(), st, st + lens - 1


class TrewHip:
    """One device context: init / submit / wait / collect."""

    def __init__(self, mode=MODE_SHORT, min_mer=5, max_mer=32, low=0.5, high=0.8, slice_length=150, device=0,
                 n_slots=2, max_batch_words=1 << 22, max_batch_reads=1 << 18, table_log2_slots=20, flags=0, max_batch_ascii_bytes=0):
        self.lib = load()
        flags |= int(os.environ.get("TREW_EXTRA_FLAGS", "0"))  # e.g. 32 = FLAG_DEBUG_POISON_LDS for a whole test run
        self.params = Params(min_mer, max_mer, low, high, slice_length, mode, device, n_slots, max_batch_words,
                             max_batch_reads, table_log2_slots, flags, max_batch_ascii_bytes)
        self.ctx = C.c_void_p()
        rc = self.lib.trew_hip_init(C.byref(self.params), C.byref(self.ctx))
        if rc != 0:
            raise TrewHipError("trew_hip_init failed (%d): %s" % (rc, self.lib.trew_hip_last_error(None).decode()))
        self.mode = mode
        self._keep = {}

    def close(self):
        if self.ctx:
            self.lib.trew_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise TrewHipError("%s failed (%d): %s" % (what, rc, self.lib.trew_hip_last_error(self.ctx).decode()))

    # ---- batches ----
    def host_batch(self, words, offsets, lengths, contiguous=False):
        """contiguous: one buffer laid out [offsets][lengths][words], which trew_hip_submit ships with a single copy."""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        if contiguous:
            n = len(offsets)
            buf = np.concatenate([offsets, lengths, words])
            b = Batch(buf.ctypes.data + 8 * n, len(words), buf.ctypes.data, buf.ctypes.data + 4 * n, 0, 0, n, 0, 0)
            b._keep = (buf,)
            return b
        b = Batch(words.ctypes.data, len(words), offsets.ctypes.data, lengths.ctypes.data, 0, 0, len(offsets), 0, 0)
        b._keep = (words, offsets, lengths)
        return b

    def device_uniform_batch(self, d_words, n_reads, read_len):
        stride = 3 * ((read_len + 31) // 32)
        return Batch(d_words, n_reads * stride, None, None, read_len, stride, n_reads, 1, read_len)

    def ascii_batch(self, reads, contiguous=True, uniform=None):
        """Text batch of the reads (bytes objects): the sequence bytes back to back + word/byte offsets + lengths, laid out
        [word_offsets][byte_offsets][lengths][bases] in one buffer when contiguous.  uniform=L: no arrays, every read has L bases."""
        reads = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        n = len(reads)
        text = np.frombuffer(b"".join(reads) + b"\0" * 8, dtype=np.uint8)
        n_bytes = len(text) - 8
        if uniform is not None:
            assert all(len(r) == uniform for r in reads)
            b = AsciiBatch(text.ctypes.data, n_bytes, None, None, None, uniform, 0, n)
            b._keep = (text,)
            return b
        lens = np.array([len(r) for r in reads], dtype=np.uint32)
        boff = np.zeros(n, dtype=np.uint32)
        woff = np.zeros(n, dtype=np.uint32)
        if n:
            boff[1:] = np.cumsum(lens[:-1], dtype=np.uint64).astype(np.uint32)
            woff[1:] = np.cumsum(3 * ((lens[:-1].astype(np.uint64) + 31) // 32)).astype(np.uint32)
        if contiguous:
            buf = np.concatenate([woff.view(np.uint8), boff.view(np.uint8), lens.view(np.uint8), text])
            base = buf.ctypes.data
            b = AsciiBatch(base + 12 * n, n_bytes, base + 4 * n, base + 8 * n, base, 0, 0, n)
            b._keep = (buf,)
            return b
        b = AsciiBatch(text.ctypes.data, n_bytes, boff.ctypes.data, lens.ctypes.data, woff.ctypes.data, 0, 0, n)
        b._keep = (text, boff, lens, woff)
        return b

    def submit_ascii(self, batch, slot=0):
        self._keep[slot] = batch
        self._chk(self.lib.trew_hip_submit_ascii(self.ctx, C.byref(batch), slot), "trew_hip_submit_ascii")

    def pack_ascii(self, batch):
        """The packed words the device makes of a text batch (diagnostic: must equal pack_reads())."""
        n = C.c_uint64(0)
        self._chk(self.lib.trew_hip_pack_ascii(self.ctx, C.byref(batch), None, 0, C.byref(n)), "trew_hip_pack_ascii")
        words = np.zeros(max(int(n.value), 1), dtype=np.uint32)
        self._chk(self.lib.trew_hip_pack_ascii(self.ctx, C.byref(batch), words.ctypes.data, len(words), C.byref(n)), "trew_hip_pack_ascii")
        return words[: int(n.value)]

    def submit(self, batch, slot=0):
        self._keep[slot] = batch
        self._chk(self.lib.trew_hip_submit(self.ctx, C.byref(batch), slot), "trew_hip_submit")

    def wait(self, slot=0):
        self._chk(self.lib.trew_hip_wait(self.ctx, slot), "trew_hip_wait")

    def submit_reads(self, reads, slot=0):
        b = self.host_batch(*pack_reads(reads))
        self.submit(b, slot)
        return b

    # ---- results ----
    def collect_rows(self, table=-1):
        """Rows of one table (or all, table=-1) as a structured numpy array (ROW_DTYPE)."""
        n = C.c_uint64(0)
        cap = getattr(self, "_collect_cap", 1 << 16)
        while True:  # one call when the guess holds; the table tells its size when it does not
            rows = np.zeros(cap, dtype=ROW_DTYPE)
            self._chk(self.lib.trew_hip_collect(self.ctx, table, C.cast(rows.ctypes.data, C.POINTER(Row)), cap,
                                                C.byref(n)), "trew_hip_collect")
            if n.value <= cap:
                return rows[: n.value]
            cap = self._collect_cap = int(n.value) + 1024

    def collect(self):
        """The six tables as {name: {(k, word): count}}."""
        return rows_to_tables(self.collect_rows(-1))

    def reset_tables(self):
        self._chk(self.lib.trew_hip_reset_tables(self.ctx), "trew_hip_reset_tables")

    def add_rows(self, tables):
        rows = tables if isinstance(tables, np.ndarray) else tables_to_rows(tables)
        rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        if len(rows):
            self._chk(self.lib.trew_hip_add_rows(self.ctx, C.cast(rows.ctypes.data, C.POINTER(Row)), len(rows)),
                      "trew_hip_add_rows")

    def collect_device(self, d_rows, cap):
        """Compact every table into the device buffer d_rows (cap rows of ROW_DTYPE.itemsize bytes); returns the row count
        (which may exceed cap: nothing is written past cap then)."""
        n = C.c_uint64(0)
        self._chk(self.lib.trew_hip_collect_device(self.ctx, d_rows, cap, C.byref(n)), "trew_hip_collect_device")
        return int(n.value)

    def collect_slice_device(self, d_slice, slice_rows, consumer_stream=None, want_count=False):
        """Compact every table into rows 1.. of the exchange slice d_slice (1 + slice_rows rows) and write its header row on
        the device; the collective the caller issues on consumer_stream is ordered behind it without a host hop.  Returns the
        row count when want_count (one host synchronisation), else None."""
        n = C.c_uint64(0)
        self._chk(self.lib.trew_hip_collect_slice_device(self.ctx, d_slice, slice_rows, consumer_stream, C.byref(n) if want_count else None),
                  "trew_hip_collect_slice_device")
        return int(n.value) if want_count else None

    def add_rows_device(self, d_rows, n_rows):
        self._chk(self.lib.trew_hip_add_rows_device(self.ctx, d_rows, n_rows), "trew_hip_add_rows_device")

    def add_gathered_device(self, d_buf, n_slices, own_slice, slice_rows, producer_stream=None):
        """One kernel over the gather buffer of the table exchange (n_slices x (1 + slice_rows) rows, headers in row 0 of each
        slice): adds every slice but own_slice.  Returns the largest header count; if it exceeds slice_rows nothing was added."""
        mx = C.c_uint64(0)
        self._chk(self.lib.trew_hip_add_gathered_device(self.ctx, d_buf, n_slices, own_slice, slice_rows, producer_stream, C.byref(mx)),
                  "trew_hip_add_gathered_device")
        return int(mx.value)

    def debug_counters(self):
        """{name: count} of the kernels' rare fall-back paths since the last reset_tables (DEBUG_COUNTERS)."""
        out = (C.c_uint64 * len(DEBUG_COUNTERS))()
        self._chk(self.lib.trew_hip_debug_counters(self.ctx, out, len(DEBUG_COUNTERS)), "trew_hip_debug_counters")
        return dict(zip(DEBUG_COUNTERS, (int(x) for x in out)))

    def debug_worklist(self, slot=0):
        """Unit indices the prefilter of the last submit on `slot` flagged (numpy uint32, worklist order)."""
        n = C.c_uint64(0)
        self._chk(self.lib.trew_hip_debug_worklist(self.ctx, slot, None, 0, C.byref(n)), "trew_hip_debug_worklist")
        out = np.zeros(int(n.value), dtype=np.uint32)
        if len(out):
            self._chk(self.lib.trew_hip_debug_worklist(self.ctx, slot, out.ctypes.data, len(out), C.byref(n)), "trew_hip_debug_worklist")
        return out

    def merge_from(self, other):
        """Add every row of `other`'s tables (another context, same or another GPU) into this context's tables."""
        self._chk(self.lib.trew_hip_merge(self.ctx, other.ctx), "trew_hip_merge")

    def table_pressure(self):
        """(used_slots, total_slots, spilled_rows, spill_capacity) -- a snapshot."""
        v = [C.c_uint64(0) for _ in range(4)]
        self._chk(self.lib.trew_hip_table_pressure(self.ctx, *[C.byref(x) for x in v]), "trew_hip_table_pressure")
        return tuple(int(x.value) for x in v)

    def segment_results(self, n_reads, slot=0):
        kh = np.zeros(n_reads, dtype=np.int32)
        kl = np.zeros(n_reads, dtype=np.int32)
        sh = np.zeros(n_reads, dtype=np.uint64)
        sl = np.zeros(n_reads, dtype=np.uint64)
        shh = np.zeros(n_reads, dtype=np.uint64)
        slh = np.zeros(n_reads, dtype=np.uint64)
        self._chk(self.lib.trew_hip_segment_results(self.ctx, slot, kh.ctypes.data, kl.ctypes.data, sh.ctypes.data,
                                                    sl.ctypes.data, shh.ctypes.data, slh.ctypes.data, n_reads),
                  "trew_hip_segment_results")
        # words as Python ints (hi << 64 | lo)
        seq_h = [(int(a) << 64) | int(b) for a, b in zip(shh, sh)]
        seq_l = [(int(a) << 64) | int(b) for a, b in zip(slh, sl)]
        return kh, kl, seq_h, seq_l

    def filter_masks(self, batch, slots_per_read):
        units = batch.n_reads // 2 if self.mode == MODE_PAIR else batch.n_reads
        cand = np.zeros(max(1, units * slots_per_read), dtype=np.uint64)
        self._chk(self.lib.trew_hip_filter_masks(self.ctx, C.byref(batch), cand.ctypes.data, slots_per_read),
                  "trew_hip_filter_masks")
        return cand[: units * slots_per_read].reshape(units, slots_per_read)

    def last_timing(self, slot=0, want_flagged=True):
        a, b, n = C.c_float(), C.c_float(), C.c_uint64()
        self._chk(self.lib.trew_hip_last_timing(self.ctx, slot, C.byref(a), C.byref(b), C.byref(n) if want_flagged else None),
                  "trew_hip_last_timing")
        return a.value, b.value, n.value

    # ---- device memory ----
    def malloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.trew_hip_malloc(self.ctx, nbytes, C.byref(p)), "trew_hip_malloc")
        return p.value

    def free(self, ptr):
        self._chk(self.lib.trew_hip_free(self.ctx, ptr), "trew_hip_free")

    def d2h(self, ptr, nbytes):
        out = np.zeros(nbytes, dtype=np.uint8)
        self._chk(self.lib.trew_hip_memcpy_d2h(self.ctx, out.ctypes.data, ptr, nbytes), "trew_hip_memcpy_d2h")
        return out

    def synth_short_device(self, seed, first_read, n_reads, read_len, d_words):
        self._chk(self.lib.trew_synth_short_device(self.ctx, seed, first_read, n_reads, read_len, d_words),
                  "trew_synth_short_device")

    def synth_long_device(self, seed, first_read, n_reads):
        """Generate long reads on the device: returns (batch, device pointers to free)."""
        lens = synth_long_lengths(seed, first_read, n_reads)
        nw = 3 * ((lens.astype(np.int64) + 31) // 32)
        offs = np.zeros(n_reads, dtype=np.int64)
        offs[1:] = np.cumsum(nw[:-1])
        total = int(offs[-1] + nw[-1]) if n_reads else 0
        if total >= 2 ** 32:
            raise TrewHipError("long batch exceeds 2^32 words")
        offs32 = offs.astype(np.uint32)
        d_words = self.malloc(total * 4 + 64)
        d_offs = self.malloc(n_reads * 4)
        d_lens = self.malloc(n_reads * 4)
        self._chk(self.lib.trew_hip_memcpy_h2d(self.ctx, d_offs, offs32.ctypes.data, n_reads * 4), "h2d")
        self._chk(self.lib.trew_hip_memcpy_h2d(self.ctx, d_lens, lens.ctypes.data, n_reads * 4), "h2d")
        self._chk(self.lib.trew_synth_long_device(self.ctx, seed, first_read, n_reads, d_offs, d_words), "trew_synth_long_device")
        b = Batch(d_words, total, d_offs, d_lens, 0, 0, n_reads, 1, int(lens.max()) if n_reads else 0)
        return b, (d_words, d_offs, d_lens), int(lens.astype(np.int64).sum())

    def synth_pair_device(self, seed, first_pair, n_pairs, read_len, d_words):
        self._chk(self.lib.trew_synth_pair_device(self.ctx, seed, first_pair, n_pairs, read_len, d_words),
                  "trew_synth_pair_device")


def k_mer_check(seq, min_mer=5, max_mer=32, low=0.5, high=0.8, flags=0, device=0):
    """k_mer_check (kmer.h:232-236) on the GPU for one segment; same result shape as the oracle's segment_check."""
    if isinstance(seq, str):
        seq = seq.encode()
    with TrewHip(mode=MODE_SEGMENT, min_mer=min_mer, max_mer=max_mer, low=low, high=high, device=device, n_slots=1,
                 max_batch_words=1 << 12, max_batch_reads=16, table_log2_slots=14, flags=flags) as t:
        t.submit_reads([seq])
        t.wait()
        kh, kl, sh, sl = t.segment_results(1)
        tabs = t.collect()
    return dict(k_high=int(kh[0]), k_low=int(kl[0]), seq_high=int(sh[0]), seq_low=int(sl[0]),
                hist_high=tabs["forward_high"], hist_low=tabs["forward_low"])
