// trew_capi.cpp -- the extern "C" layer of libtrew_hip.so (see include/trew_hip.h).
//
// Host side only: owns the HIP streams, the per-slot device buffers, the device
// count table, and launches the kernels of trew_kernels.hip.  There is no CPU
// fallback: every compute entry point fails loudly when HIP is unavailable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "trew_common.hpp"
#include "trew_launch.hpp"
#include "trew_synth.hpp"

using namespace trew;

namespace {

struct Slot {
    hipStream_t stream = nullptr;
    u32 *d_buf = nullptr;  // one allocation: [offsets max_batch_reads][lengths max_batch_reads][words max_batch_words + slack]
    u32 *d_words = nullptr, *d_offsets = nullptr, *d_lengths = nullptr;  // views into d_buf
    unsigned char *d_ascii = nullptr;  // text batches: [word_offsets][byte_offsets][lengths][bases], max_batch_ascii_bytes + slack
    u32 *d_wl = nullptr;
    u32 *d_wl_count = nullptr;
    int2 *d_thr = nullptr;   // pass thresholds of the prefilter's uniform-geometry path (kThrRows * kThrRow)
    int2 *h_thr = nullptr;   // pinned staging of the same
    hipEvent_t ev_tail = nullptr, ev_copied = nullptr;  // order the context's copy stream behind / in front of this slot's stream (BatchCopy)
    u32 *h_seen = nullptr;   // TREW_FLAG_TRACK_PRESSURE: the counter line as of the end of this slot's last batch (pinned, behind h_thr)
    u32 thr_length = 0;      // uniform read length d_thr was computed for (0: none yet)
    SegResults res = {nullptr, nullptr, nullptr, nullptr};
    // ring of (before filter, between, after exact) events: submits may be queued back to back
    // on the slot's stream without a wait in between, each keeps its own timestamps
    static constexpr int kRing = 128;
    hipEvent_t ev[kRing][3] = {};
    u64 n_launches = 0;  // submits that launched kernels: picks the counter block
    u64 n_submits = 0;   // submits since init
    u64 n_reported = 0;  // submits already averaged by trew_hip_last_timing
    u64 n_units = 0;
};

thread_local std::string g_init_error;  // trew_hip_init failures before a context exists (read back on the same thread)
// Last error of the calling thread.  Several host threads share one context (one slot each); an error string
// inside the context would be written and read concurrently, so the text lives with the thread that got the
// failing status -- the only one that reads it back.
thread_local std::string g_thread_error;
// One counter block: [0] worklist size, then 8 queue heads on separate 128-B lines.  A slot owns TWO blocks and
// alternates between them; the exact kernel of one submit clears the block of the next (see exact_kernel), so a
// submit needs no memset call of its own.  Both are cleared once at init.
constexpr size_t kWlCountWords = 32 + 8 * 32;
constexpr size_t kWlCountBytes = kWlCountWords * 4;

}  // namespace

struct trew_hip_ctx {
    trew_hip_params p;
    DevParams dp;
    DevTable table;
    DevTableG1 table_g1;  // `table` (with TREW_FLAG_DEBUG_NO_EMIT applied) + g1: the exact kernels' descriptor argument
    // TREW_FLAG_COMPAT_G1 (kernels/g1_compat.inc): the stale-row log, the pair flags and the two carry buffers (read / written
    // by a batch, swapped after it)
    DevG1 g1 = {nullptr, nullptr, nullptr, 0};
    trew_hip_row *g1_carry[2] = {nullptr, nullptr};
    u32 g1_carry_cap = 0;
    u64 g1_batches = 0;
    DevWide wide;  // host copy of *table.wide
    u64 table_slots = 0;
    std::vector<Slot> slots;
    // Every host-to-device copy of a batch goes through this ONE stream, whatever slot it is for.  On ROCm 7 the first DMA-engine
    // copy that a stream issues costs 3.5 ms of host time under a process-wide lock (measured: 30 slots = 0.105 s of a 0.22 s
    // file, the workers of the `trew` host queueing up behind each other, profiles/r03/README.md); after that a 16 MB copy is
    // queued in microseconds.  The link carries one copy at a time anyway, so one stream loses no bandwidth.
    hipStream_t copy_stream = nullptr;
    std::mutex copy_mu;  // a batch's copies and the event behind them are queued as one unit
    // Fills of device memory (the tables at init and reset, the buffers at init) are queued on this stream and every other stream
    // of the context is made to wait for the event behind them -- on the device.  hipMemset on the null stream returns at once
    // (2 us for 256 MiB, tools/memset_probe.hip) while the fill takes 0.13 ms, and the slots' non-blocking streams do not wait for
    // the null stream: a batch submitted right behind it scanned into a table that was still being cleared (round 3's "empty
    // tables once in 270 runs").  Stream order instead of a host wait: init / reset do not stall the submitting threads.
    // (The fill stream IS the copy stream: a stream of its own changed which hardware queue the slots' streams landed on -- ROCm
    // spreads a process's streams over four queues in creation order -- and the two batch slots of bench.py stopped overlapping:
    // 1.10 ms a step instead of 0.91.  Fills only happen while no batch copy is in flight.)
    hipStream_t fill_stream = nullptr;  // alias of copy_stream
    hipEvent_t ev_filled = nullptr;
    int n_cu = 256;
    // persistent scratch of trew_hip_collect (device-side compaction)
    unsigned long long *d_collect_n = nullptr;
    trew_hip_row *d_collect_rows = nullptr;
    u64 collect_cap = 0;
    // persistent scratch of trew_hip_add_rows / trew_hip_merge (grow-only)
    trew_hip_row *d_add_rows = nullptr;
    u64 add_cap = 0;
    u32 *d_row_flags = nullptr;  // kRowFlagWords words: verdict of the validation pass of the row-adding entry points
    hipEvent_t ev_producer = nullptr;  // orders slot 0's stream behind a caller's stream (trew_hip_add_gathered_device)
    std::mutex table_mu;  // collect / reset / add_rows / merge are whole-table operations: one at a time
    // TREW_FLAG_TRACK_PRESSURE: the largest value of each fill counter any host thread has read so far (they only grow
    // between resets); trew_hip_table_pressure answers from these
    std::atomic<u32> seen[kDiagWords];
    std::atomic<int> last_slot{-1};  // slot of the most recent submit (launch_batch: is another slot's batch still running?)
    std::mutex seen_mu;  // keeps a reset (which zeroes the copies) apart from a query that folds them
};

// fold one copy of the counter line into ctx->seen
static void note_seen(trew_hip_ctx *ctx, const u32 *diag) {
    for (int i : {(int) kDiagOverflow, (int) kDiagInserted, (int) kDiagInsertedWide, (int) kDiagSpillRows}) {
        u32 cur = ctx->seen[i].load(std::memory_order_relaxed);
        while (cur < diag[i] && !ctx->seen[i].compare_exchange_weak(cur, diag[i], std::memory_order_relaxed)) {
        }
    }
}

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (void) (ctx);                                                                         \
            g_thread_error = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            return (int) e_ ? (int) e_ : -1;                                                      \
        }                                                                                         \
    } while (0)

static int fail(trew_hip_ctx *, const std::string &msg) {
    g_thread_error = msg;
    return -1;
}

extern "C" int trew_hip_abi_version(void) { return TREW_HIP_ABI_VERSION; }

extern "C" int trew_hip_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int trew_hip_host_alloc(trew_hip_ctx *ctx, uint64_t bytes, void **h_ptr) {
    if (!ctx || !h_ptr) return -1;
    hipError_t e = hipSetDevice(ctx->p.device);
    if (e == hipSuccess) e = hipHostMalloc(h_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        g_thread_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return (int) e;
    }
    return 0;
}

extern "C" int trew_hip_host_free(trew_hip_ctx *ctx, void *h_ptr) {
    if (!ctx) return -1;
    hipError_t e = hipHostFree(h_ptr);
    if (e != hipSuccess) {
        g_thread_error = std::string("hipHostFree: ") + hipGetErrorString(e);
        return (int) e;
    }
    return 0;
}

extern "C" const char *trew_hip_last_error(const trew_hip_ctx *ctx) {
    return ctx ? g_thread_error.c_str() : g_init_error.c_str();
}

static float conservative_lowf(double low) {
    float f = (float) (low * (1.0 - 1e-6));
    f = std::nextafterf(f, 0.0f);
    return f;
}

// an event behind everything queued on the fill stream so far; the copy stream and every slot's stream wait for it on the device
static int order_behind_fills(trew_hip_ctx *ctx) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_filled, ctx->fill_stream));
    for (auto &s : ctx->slots)
        if (s.stream) HIPCHK(ctx, hipStreamWaitEvent(s.stream, ctx->ev_filled, 0));
    return 0;
}

extern "C" int trew_hip_init(const trew_hip_params *params, trew_hip_ctx **out) {
    if (!params || !out) {
        g_init_error = "trew_hip_init: null argument";
        return -1;
    }
    *out = nullptr;
    const trew_hip_params &p = *params;
    // same limits as the reference CLI (trew.cpp:175-228, 256-304) + this ABI version's device limits
    if (p.min_mer > p.max_mer) { g_init_error = "MIN_MER must not be greater than MAX_MER."; return -1; }
    if (p.min_mer < 3) { g_init_error = "MIN_MER must be greater than or equal to 3."; return -1; }
    if (p.max_mer > 64) { g_init_error = "MAX_MER must be less than or equal to 64."; return -1; }
    if (!(0 < p.low_baseline && p.low_baseline <= 1) || !(0 < p.high_baseline && p.high_baseline <= 1)) { g_init_error = "Baseline must be in range 0 to 1."; return -1; }
    if (p.low_baseline > p.high_baseline) { g_init_error = "Low baseline must be smaller than high baseline."; return -1; }
    if (p.mode < TREW_MODE_SHORT || p.mode > TREW_MODE_SEGMENT) { g_init_error = "unknown mode"; return -1; }
    if (p.mode == TREW_MODE_LONG) {
        if (p.slice_length < 2 * p.max_mer) { g_init_error = "SLICE_LENGTH must be greater than or equal to twice of MAX_MER."; return -1; }
        if (2 * p.slice_length - 1 > kMaxSegBases) { g_init_error = "SLICE_LENGTH must be at most 512 on the HIP path."; return -1; }
    }
    if ((p.flags & TREW_FLAG_COMPAT_G1) && (p.mode != TREW_MODE_PAIR || p.max_mer > 32 || p.n_slots != 1)) {
        g_init_error = "TREW_FLAG_COMPAT_G1 needs pair mode, MAX_MER <= 32 (the 128-bit branch has no stale map) and n_slots = 1 (file order)";
        return -1;
    }
    if (p.n_slots < 1 || p.n_slots > 512) { g_init_error = "n_slots must be in [1,512]"; return -1; }
    if (p.table_log2_slots < 12 || p.table_log2_slots > 30) { g_init_error = "table_log2_slots must be in [12,30]"; return -1; }
    if (p.max_batch_reads == 0 || p.max_batch_reads > 0xfffffff0ull) { g_init_error = "max_batch_reads out of range"; return -1; }

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_init_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return -2;
    }
    if (p.device < 0 || p.device >= ndev) { g_init_error = "device ordinal out of range"; return -1; }
    e = hipSetDevice(p.device);
    if (e != hipSuccess) { g_init_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return -2; }

    trew_hip_ctx *ctx = new trew_hip_ctx();
    for (auto &v : ctx->seen) v.store(0, std::memory_order_relaxed);
    memset(&ctx->table, 0, sizeof(ctx->table));
    memset(&ctx->wide, 0, sizeof(ctx->wide));
    ctx->p = p;
    ctx->dp.min_mer = p.min_mer;
    ctx->dp.max_mer = p.max_mer;
    ctx->dp.low = p.low_baseline;
    ctx->dp.high = p.high_baseline;
    ctx->dp.lowf = conservative_lowf(p.low_baseline);
    ctx->dp.slice_len = p.slice_length > 0 ? p.slice_length : 150;
    ctx->dp.mode = p.mode;
    ctx->dp.flags = p.flags;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p.device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;

    auto bail = [&](const char *what, hipError_t er) {
        g_init_error = std::string(what) + ": " + hipGetErrorString(er);
        trew_hip_destroy(ctx);
        return -3;
    };
    if ((e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    ctx->fill_stream = ctx->copy_stream;
    if ((e = hipEventCreateWithFlags(&ctx->ev_filled, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    ctx->table_slots = 1ull << p.table_log2_slots;
    ctx->table.log2_part_slots = p.table_log2_slots - kTablePartBits;
    if ((e = hipMalloc((void **) &ctx->table.keys, ctx->table_slots * 8)) != hipSuccess) return bail("hipMalloc(table keys)", e);
    if ((e = hipMalloc((void **) &ctx->table.counts, ctx->table_slots * 8)) != hipSuccess) return bail("hipMalloc(table counts)", e);
    if ((e = hipMalloc((void **) &ctx->table.overflow, kDiagWords * 4)) != hipSuccess) return bail("hipMalloc(overflow)", e);
    if ((e = hipMemsetAsync(ctx->table.keys, 0, ctx->table_slots * 8, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
    if ((e = hipMemsetAsync(ctx->table.counts, 0, ctx->table_slots * 8, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
    if ((e = hipMemsetAsync(ctx->table.overflow, 0, kDiagWords * 4, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
    // wide entries (k > 32) are rare: a quarter of the slots
    ctx->wide.wide_log2_slots = std::max<u32>(10u, p.table_log2_slots - 2u);
    {
        const size_t wb = (size_t) 8 << ctx->wide.wide_log2_slots;
        u64 **arr[4] = {&ctx->wide.wtag, &ctx->wide.wlo, &ctx->wide.whi, &ctx->wide.wcount};
        for (auto a : arr) {
            if ((e = hipMalloc((void **) a, wb)) != hipSuccess) return bail("hipMalloc(wide table)", e);
            if ((e = hipMemsetAsync(*a, 0, wb, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
        }
        // spill log: 1/16 of the table's slots, at least 64 k rows
        ctx->wide.spill_cap = (u32) std::max<u64>(1ull << 16, ctx->table_slots >> 4);
        ctx->wide.spin_limit = (p.flags & TREW_FLAG_DEBUG_WIDE_NO_WAIT) ? 0u : (1u << 20);
        if ((e = hipMalloc((void **) &ctx->wide.spill_rows, (size_t) ctx->wide.spill_cap * sizeof(trew_hip_row))) != hipSuccess) return bail("hipMalloc(spill log)", e);
        ctx->wide.spill_n = ctx->table.overflow + kDiagSpillRows;  // cleared with the counter line
        DevWide *dw = nullptr;
        if ((e = hipMalloc((void **) &dw, sizeof(DevWide))) != hipSuccess) return bail("hipMalloc(wide descriptor)", e);
        if ((e = hipMemcpy(dw, &ctx->wide, sizeof(DevWide), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
        ctx->table.wide = dw;
    }
    {
        if (p.flags & TREW_FLAG_COMPAT_G1) {
            // a pair's whole-read block records the classes of at most two reads for two baselines; the log is sized for
            // every pair of a full batch doing so with a dozen classes each and reports (never hides) an overflow
            ctx->g1.log_cap = (u32) std::min<u64>(std::max<u64>(1ull << 16, 24ull * (p.max_batch_reads / 2)), 1ull << 26);
            ctx->g1_carry_cap = 1u << 16;
            if ((e = hipMalloc((void **) &ctx->g1.log, (size_t) ctx->g1.log_cap * sizeof(trew_hip_row))) != hipSuccess) return bail("hipMalloc(G1 log)", e);
            if ((e = hipMalloc((void **) &ctx->g1.counters, 16)) != hipSuccess) return bail("hipMalloc(G1 counters)", e);
            if ((e = hipMemsetAsync(ctx->g1.counters, 0, 16, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
            if ((e = hipMalloc((void **) &ctx->g1.pair_flags, (size_t) (p.max_batch_reads / 2 + 64))) != hipSuccess) return bail("hipMalloc(G1 pair flags)", e);
            for (auto &cb : ctx->g1_carry)
                if ((e = hipMalloc((void **) &cb, (size_t) ctx->g1_carry_cap * sizeof(trew_hip_row))) != hipSuccess) return bail("hipMalloc(G1 carry)", e);
        }
        DevTableG1 tbl;
        tbl.t = ctx->table;
        tbl.g = ctx->g1;
        tbl.w = ctx->wide;
        if (p.flags & TREW_FLAG_DEBUG_NO_EMIT) tbl.t.log2_part_slots = 0xffffffffu;  // cached_add drops every row
        ctx->table_g1 = tbl;
    }

    if ((e = hipMalloc((void **) &ctx->d_row_flags, kRowFlagWords * 4)) != hipSuccess) return bail("hipMalloc(row flags)", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_producer, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    ctx->slots.resize((size_t) p.n_slots);
    for (auto &s : ctx->slots) {
        if ((e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
        if ((e = hipEventCreateWithFlags(&s.ev_tail, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&s.ev_copied, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        {
            // +8 words of slack: the exact kernel fetches whole 64-word heads of a read
            const size_t words = 2 * (size_t) p.max_batch_reads + (size_t) p.max_batch_words + 8;
            if ((e = hipMalloc((void **) &s.d_buf, words * 4)) != hipSuccess) return bail("hipMalloc(batch buffer)", e);
            if ((e = hipMemsetAsync(s.d_buf, 0, words * 4, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
            s.d_offsets = s.d_buf;
            s.d_lengths = s.d_buf + p.max_batch_reads;
            s.d_words = s.d_buf + 2 * (size_t) p.max_batch_reads;
        }
        if (p.max_batch_ascii_bytes) {
            if ((e = hipMalloc((void **) &s.d_ascii, p.max_batch_ascii_bytes + 256)) != hipSuccess) return bail("hipMalloc(text buffer)", e);
            if ((e = hipMemsetAsync(s.d_ascii, 0, p.max_batch_ascii_bytes + 256, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
        }
        if ((e = hipMalloc((void **) &s.d_thr, kThrTableBytes)) != hipSuccess) return bail("hipMalloc(thresholds)", e);
        if ((e = hipHostMalloc((void **) &s.h_thr, kThrTableBytes + kDiagWords * 4, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc(thresholds)", e);
        s.h_seen = (u32 *) ((char *) s.h_thr + kThrTableBytes);
        memset(s.h_seen, 0, kDiagWords * 4);
        if ((e = hipMalloc((void **) &s.d_wl, p.max_batch_reads * sizeof(u32))) != hipSuccess) return bail("hipMalloc(worklist)", e);
        if ((e = hipMalloc((void **) &s.d_wl_count, 2 * kWlCountBytes)) != hipSuccess) return bail("hipMalloc(wl_count)", e);
        if ((e = hipMemsetAsync(s.d_wl_count, 0, 2 * kWlCountBytes, ctx->fill_stream)) != hipSuccess) return bail("hipMemsetAsync", e);
        if (p.mode == TREW_MODE_SEGMENT) {
            if ((e = hipMalloc((void **) &s.res.k_high, p.max_batch_reads * 4)) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc((void **) &s.res.k_low, p.max_batch_reads * 4)) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc((void **) &s.res.seq_high, p.max_batch_reads * 8)) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc((void **) &s.res.seq_low, p.max_batch_reads * 8)) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc((void **) &s.res.seq_high_hi, p.max_batch_reads * 8)) != hipSuccess) return bail("hipMalloc", e);
            if ((e = hipMalloc((void **) &s.res.seq_low_hi, p.max_batch_reads * 8)) != hipSuccess) return bail("hipMalloc", e);
        }
        for (int i = 0; i < Slot::kRing; i++)
            for (int j = 0; j < 3; j++)
                if ((e = hipEventCreate(&s.ev[i][j])) != hipSuccess) return bail("hipEventCreate", e);
    }
    // every stream of the context starts behind the fills (see fill_stream)
    if (int rc = order_behind_fills(ctx)) {
        g_init_error = g_thread_error;
        trew_hip_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return 0;
}

extern "C" void trew_hip_destroy(trew_hip_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->p.device);
    for (auto &s : ctx->slots) {
        if (s.stream) (void) hipStreamSynchronize(s.stream);
        if (s.d_buf) (void) hipFree(s.d_buf);
        if (s.d_ascii) (void) hipFree(s.d_ascii);
        if (s.d_thr) (void) hipFree(s.d_thr);
        if (s.h_thr) (void) hipHostFree(s.h_thr);
        if (s.d_wl) (void) hipFree(s.d_wl);
        if (s.d_wl_count) (void) hipFree(s.d_wl_count);
        if (s.res.k_high) (void) hipFree(s.res.k_high);
        if (s.res.k_low) (void) hipFree(s.res.k_low);
        if (s.res.seq_high) (void) hipFree(s.res.seq_high);
        if (s.res.seq_low) (void) hipFree(s.res.seq_low);
        if (s.res.seq_high_hi) (void) hipFree(s.res.seq_high_hi);
        if (s.res.seq_low_hi) (void) hipFree(s.res.seq_low_hi);
        for (int i = 0; i < Slot::kRing; i++)
            for (int j = 0; j < 3; j++)
                if (s.ev[i][j]) (void) hipEventDestroy(s.ev[i][j]);
        if (s.ev_tail) (void) hipEventDestroy(s.ev_tail);
        if (s.ev_copied) (void) hipEventDestroy(s.ev_copied);
        if (s.stream) (void) hipStreamDestroy(s.stream);
    }
    if (ctx->copy_stream) (void) hipStreamSynchronize(ctx->copy_stream);
    if (ctx->copy_stream) (void) hipStreamDestroy(ctx->copy_stream);
    if (ctx->ev_filled) (void) hipEventDestroy(ctx->ev_filled);
    if (ctx->table.keys) (void) hipFree(ctx->table.keys);
    if (ctx->table.counts) (void) hipFree(ctx->table.counts);
    if (ctx->table.overflow) (void) hipFree(ctx->table.overflow);
    if (ctx->wide.wtag) (void) hipFree(ctx->wide.wtag);
    if (ctx->wide.wlo) (void) hipFree(ctx->wide.wlo);
    if (ctx->wide.whi) (void) hipFree(ctx->wide.whi);
    if (ctx->wide.wcount) (void) hipFree(ctx->wide.wcount);
    if (ctx->wide.spill_rows) (void) hipFree(ctx->wide.spill_rows);
    if (ctx->table.wide) (void) hipFree((void *) ctx->table.wide);
    if (ctx->g1.log) (void) hipFree(ctx->g1.log);
    if (ctx->g1.counters) (void) hipFree(ctx->g1.counters);
    if (ctx->g1.pair_flags) (void) hipFree(ctx->g1.pair_flags);
    for (auto cb : ctx->g1_carry)
        if (cb) (void) hipFree(cb);
    if (ctx->d_collect_n) (void) hipFree(ctx->d_collect_n);
    if (ctx->d_collect_rows) (void) hipFree(ctx->d_collect_rows);
    if (ctx->d_add_rows) (void) hipFree(ctx->d_add_rows);
    if (ctx->d_row_flags) (void) hipFree(ctx->d_row_flags);
    if (ctx->ev_producer) (void) hipEventDestroy(ctx->ev_producer);
    delete ctx;
}

// Longest segment any slot of any read of the batch can have, and validation of
// the read-length limits (short mode aborts above MAX_SEQ = 1000, kmer.cpp:1006-1009;
// the build applies the same limit to pair mode).
static int batch_geometry(trew_hip_ctx *ctx, const trew_hip_batch *b, u32 *max_seg, u32 *max_len) {
    const int mode = ctx->p.mode;
    const int MINM = ctx->p.min_mer, MAXM = ctx->p.max_mer, SL = ctx->dp.slice_len;
    u32 maxlen = 0, ms = 0;
    // longest segment of one unit, from the same geometry function the kernels use
    auto unit_seg = [&](u32 n1, u32 n2) {
        u32 m = 0;
        // long mode also walks the interior slices: the middle one carries the remainder (kmer.cpp:790-798)
        if (mode == TREW_MODE_LONG && (int) n1 >= SL) m = (u32) SL + n1 % (u32) SL;
        for (int slot = 0; slot < mode_slots(mode); slot++) {
            const Segment sg = get_segment(mode, slot, n1, n2, MINM, MAXM, SL);
            if (sg.valid) m = std::max(m, sg.len);
        }
        return m;
    };
    if (b->lengths && !b->on_device) {
        // Longest segment over the batch.  It only sizes the kernels (mask words, LDS), so a value that is never
        // too small is what matters; the closed forms below equal unit_seg() except that a whole-read segment is
        // assumed whenever the read is shorter than 4*MAX_MER (get_segment also wants kmin <= kmax).  One pass of
        // a few instructions per read instead of three get_segment() calls: the CLI submits ~10^4 batches a second.
        const u32 two_min = (u32) (2 * MINM), four_min = (u32) (4 * MINM), four_max = (u32) (4 * MAXM);
        if (mode == TREW_MODE_PAIR) {
            for (u64 i = 0; i + 1 < b->n_reads; i += 2) {
                const u32 a = b->lengths[i], c = b->lengths[i + 1], hi = std::max(a, c), lo = std::min(a, c);
                maxlen = std::max(maxlen, hi);
                if (lo >= four_min) ms = std::max(ms, (hi + 1) / 2);
                if (lo >= two_min && lo < four_max) ms = std::max(ms, hi);
            }
        } else if (mode == TREW_MODE_SHORT) {
            for (u64 i = 0; i < b->n_reads; i++) {
                const u32 n = b->lengths[i];
                maxlen = std::max(maxlen, n);
                if (n >= two_min && n < four_max) ms = std::max(ms, n);
            }
            if (maxlen >= four_min) ms = std::max(ms, (maxlen + 1) / 2);
        } else if (mode == TREW_MODE_LONG) {
            for (u64 i = 0; i < b->n_reads; i++) {
                const u32 n = b->lengths[i];
                maxlen = std::max(maxlen, n);
                if ((int) n >= SL) ms = std::max(ms, (u32) SL + n % (u32) SL);  // the middle slice carries the remainder (kmer.cpp:790-798)
            }
        } else {
            for (u64 i = 0; i < b->n_reads; i++) {
                maxlen = std::max(maxlen, b->lengths[i]);
                ms = std::max(ms, unit_seg(b->lengths[i], 0));
            }
        }
    } else if (b->lengths) {
        // device-resident ragged batch: only the caller's max_length hint is known
        maxlen = b->max_length > 0 ? (u32) b->max_length : (u32) kMaxSegBases;
        if (mode == TREW_MODE_SHORT || mode == TREW_MODE_PAIR)
            ms = std::max((maxlen + 1) / 2, std::min<u32>(maxlen, (u32) (4 * MAXM - 1)));
        else if (mode == TREW_MODE_LONG)
            ms = std::min<u32>(maxlen, (u32) (2 * SL - 1));
        else
            ms = maxlen;
    } else {
        maxlen = b->uniform_length;
        ms = unit_seg(maxlen, maxlen);
    }
    // short mode aborts above MAX_SEQ = 1000 (kmer.cpp:1006-1009); the build applies the
    // same limit to pair mode, which the reference leaves unchecked
    if ((mode == TREW_MODE_SHORT || mode == TREW_MODE_PAIR) && maxlen > 1000)
        return fail(ctx, "This mode is designed for short-read sequencing. Please use 'trew long'.");
    if (mode == TREW_MODE_SEGMENT && maxlen > (u32) kMaxSegBases) return fail(ctx, "segment longer than 1023 bases");
    *max_seg = ms;
    *max_len = maxlen;
    return 0;
}

// The host-to-device copies of one batch: queued on the context's copy stream (see trew_hip_ctx::copy_stream), behind whatever
// the slot's stream still has to do with the slot's device buffers and in front of the kernels that follow on it.
struct BatchCopy {
    trew_hip_ctx *ctx;
    Slot &s;
    std::unique_lock<std::mutex> lk;
    BatchCopy(trew_hip_ctx *c, Slot &sl) : ctx(c), s(sl) {}
    int begin() {
        lk = std::unique_lock<std::mutex>(ctx->copy_mu);
        if (hipStreamQuery(s.stream) != hipSuccess) {  // back-to-back submits on one slot: its buffers are still being read
            (void) hipGetLastError();                  // hipErrorNotReady is not an error
            HIPCHK(ctx, hipEventRecord(s.ev_tail, s.stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, s.ev_tail, 0));
        }
        return 0;
    }
    int copy(void *dst, const void *src, u64 bytes) {
        HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
        return 0;
    }
    int end() {
        HIPCHK(ctx, hipEventRecord(s.ev_copied, ctx->copy_stream));
        lk.unlock();
        HIPCHK(ctx, hipStreamWaitEvent(s.stream, s.ev_copied, 0));
        return 0;
    }
};

static int stage_batch(trew_hip_ctx *ctx, const trew_hip_batch *b, Slot &s, DevBatch *db) {
    if (b->n_reads > ctx->p.max_batch_reads) return fail(ctx, "batch has more reads than max_batch_reads");
    if (ctx->p.mode == TREW_MODE_PAIR && (b->n_reads & 1)) return fail(ctx, "pair mode needs an even number of reads");
    if (!b->offsets && !b->lengths && b->uniform_stride < 3 * ((b->uniform_length + 31) / 32))
        return fail(ctx, "uniform_stride smaller than the packed read");
    if ((b->offsets == nullptr) != (b->lengths == nullptr)) return fail(ctx, "offsets and lengths must both be given or both be NULL");
    // Index widths of the kernels: worklist entries are u32 unit indices, a read starts at a u32 word offset with explicit
    // offsets and at unit * stride (64-bit arithmetic, get_read) without.  max_batch_reads <= 0xfffffff0 bounds the first for
    // every batch; a uniform batch (device-resident ones included, where no n_words check applies) must also keep the
    // word index of its last read inside what one allocation can hold.
    if (b->n_reads > 0xfffffff0ull) return fail(ctx, "batch has more reads than the kernels' 32-bit unit index holds");
    if (!b->offsets && b->n_reads && (b->n_reads - 1) > (0xffffffffffffffffull / 4ull - 4096ull) / std::max<u64>(1, b->uniform_stride))
        return fail(ctx, "uniform batch: n_reads * uniform_stride overflows the 64-bit word index");
    db->uniform_length = b->uniform_length;
    db->uniform_stride = b->uniform_stride;
    db->n_reads = b->n_reads;
    db->n_units = ctx->p.mode == TREW_MODE_PAIR ? b->n_reads / 2 : b->n_reads;
    if (b->on_device) {
        db->words = b->words;
        db->offsets = b->offsets;
        db->lengths = b->lengths;
    } else {
        if (b->n_words > ctx->p.max_batch_words) return fail(ctx, "batch has more words than max_batch_words");
        if (b->offsets) {
            // the kernels index words[] with these: a read that points outside the batch must never reach the device
            for (u64 i = 0; i < b->n_reads; i++)
                if ((u64) b->offsets[i] + 3ull * (((u64) b->lengths[i] + 31ull) / 32ull) > b->n_words) return fail(ctx, "a read's offset / length points outside the batch's words");
        } else if (b->n_reads && (b->n_reads - 1) * (u64) b->uniform_stride + 3ull * (((u64) b->uniform_length + 31ull) / 32ull) > b->n_words) {
            return fail(ctx, "uniform batch: n_reads * stride exceeds the batch's words");
        }
        // The caller laid the three arrays out back to back in one buffer (see trew_hip.h): ONE copy.  A host that
        // submits thousands of batches a second is bound by HIP API calls, not by bytes.
        BatchCopy bc(ctx, s);
        if (int rc = bc.begin()) return rc;
        if (b->offsets && b->lengths == b->offsets + b->n_reads && b->words == b->lengths + b->n_reads) {  // [offsets][lengths][words]
            if (int rc = bc.copy(s.d_buf, b->offsets, (2 * b->n_reads + b->n_words) * 4)) return rc;
            if (int rc = bc.end()) return rc;
            db->offsets = s.d_buf;
            db->lengths = s.d_buf + b->n_reads;
            db->words = s.d_buf + 2 * b->n_reads;
            return 0;
        }
        if (b->offsets && b->offsets == b->words + b->n_words && b->lengths == b->offsets + b->n_reads) {  // [words][offsets][lengths]
            if (int rc = bc.copy(s.d_buf, b->words, (2 * b->n_reads + b->n_words) * 4)) return rc;
            if (int rc = bc.end()) return rc;
            db->words = s.d_buf;
            db->offsets = s.d_buf + b->n_words;
            db->lengths = s.d_buf + b->n_words + b->n_reads;
            return 0;
        }
        if (int rc = bc.copy(s.d_words, b->words, b->n_words * 4)) return rc;
        db->words = s.d_words;
        db->offsets = nullptr;
        db->lengths = nullptr;
        if (b->offsets) {
            if (int rc = bc.copy(s.d_offsets, b->offsets, b->n_reads * 4)) return rc;
            if (int rc = bc.copy(s.d_lengths, b->lengths, b->n_reads * 4)) return rc;
            db->offsets = s.d_offsets;
            db->lengths = s.d_lengths;
        }
        if (int rc = bc.end()) return rc;
    }
    return 0;
}

// Thresholds of the prefilter's uniform-geometry path for this batch (nullptr: the batch is ragged).  They depend only
// on the read length, so a slot recomputes and re-sends its 3 KB table when the length changes; the copy is queued on
// the slot's stream, in front of the kernel that reads it.
static int stage_thresholds(trew_hip_ctx *ctx, Slot &s, const DevBatch &db, const int2 **out) {
    *out = nullptr;
    if (db.offsets || db.lengths || db.uniform_length == 0) return 0;
    if (s.thr_length != db.uniform_length) {
        HIPCHK(ctx, hipStreamSynchronize(s.stream));  // an earlier copy may still be reading the staging buffer
        fill_thresholds(ctx->dp, db.uniform_length, s.h_thr);
        HIPCHK(ctx, hipMemcpyAsync(s.d_thr, s.h_thr, kThrTableBytes, hipMemcpyHostToDevice, s.stream));
        s.thr_length = db.uniform_length;
    }
    *out = s.d_thr;
    return 0;
}

static int sync_all(trew_hip_ctx *ctx);

// the two kernels of one staged batch on the slot's stream (everything trew_hip_submit does after the copy)
static int launch_batch(trew_hip_ctx *ctx, Slot &s, const DevBatch &db, u32 max_seg, u32 max_len) {
    s.n_units = db.n_units;
    if (db.n_units == 0) return 0;
    if (ctx->p.mode == TREW_MODE_SEGMENT) {
        HIPCHK(ctx, hipMemsetAsync(s.res.k_high, 0, db.n_reads * 4, s.stream));
        HIPCHK(ctx, hipMemsetAsync(s.res.k_low, 0, db.n_reads * 4, s.stream));
        HIPCHK(ctx, hipMemsetAsync(s.res.seq_high, 0, db.n_reads * 8, s.stream));
        HIPCHK(ctx, hipMemsetAsync(s.res.seq_low, 0, db.n_reads * 8, s.stream));
        HIPCHK(ctx, hipMemsetAsync(s.res.seq_high_hi, 0, db.n_reads * 8, s.stream));
        HIPCHK(ctx, hipMemsetAsync(s.res.seq_low_hi, 0, db.n_reads * 8, s.stream));
    }
    const u32 wl_cap = (u32) ctx->p.max_batch_reads;
    // Counter block of this launch: clean because the exact kernel of the previous successful launch cleared it (or init did).
    // n_launches only advances once BOTH kernels are queued: if anything below fails, the block may hold a half-built worklist
    // and no exact kernel will clear the other one, so the failing path wipes both before returning (submit_failed).
    u32 *const wl_count = s.d_wl_count + (s.n_launches & 1) * kWlCountWords;
    u32 *const wl_count_next = s.d_wl_count + ((s.n_launches + 1) & 1) * kWlCountWords;  // this launch clears it
    auto submit_failed = [&](int rc) {
        const std::string keep = g_thread_error;  // the clean-up must not replace the text of the failure
        (void) hipMemsetAsync(s.d_wl_count, 0, 2 * kWlCountBytes, s.stream);
        (void) hipStreamSynchronize(s.stream);
        g_thread_error = keep;
        return rc;
    };
#define SUBMIT_CHK(expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            g_thread_error = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return submit_failed((int) e_);                                                    \
        }                                                                                      \
    } while (0)
    const int2 *d_thr = nullptr;
    if (int rc = stage_thresholds(ctx, s, db, &d_thr)) return submit_failed(rc);
    const bool compat_g1 = (ctx->p.flags & TREW_FLAG_COMPAT_G1) != 0;
    if (compat_g1) SUBMIT_CHK(hipMemsetAsync(ctx->g1.pair_flags, 0, db.n_units, s.stream));
    const bool timed = !(ctx->p.flags & TREW_FLAG_NO_TIMING);
    hipEvent_t *ev = s.ev[s.n_submits % Slot::kRing];
    if (timed) SUBMIT_CHK(hipEventRecord(ev[0], s.stream));
    SUBMIT_CHK(launch_filter(s.stream, (u32) ctx->n_cu, max_seg, ctx->dp, db, s.d_wl, wl_count, wl_cap, nullptr, 0, ctx->table.overflow, d_thr));
    if (timed) SUBMIT_CHK(hipEventRecord(ev[1], s.stream));
    // LDS working set of the exact kernel: the longest segment it may stage (the whole
    // read for k_mer_target / the whole-read check; a slice pair in long mode)
    const u32 exact_seg = ctx->p.mode == TREW_MODE_LONG ? std::min<u32>(max_len, (u32) (2 * ctx->dp.slice_len - 1)) : max_len;
    const u32 cap = std::max<u32>(64u, ((exact_seg + 1 + 63u) / 64u) * 64u);
    const u32 rawwords = ctx->p.mode == TREW_MODE_LONG ? 4u : 3u * ((max_len + 31u) / 32u) + 1u;
    // is the previous submit (on another slot) still running?  Then this batch's kernels will share the chip with it.
    const int me = (int) (&s - ctx->slots.data());
    const int prev = ctx->last_slot.exchange(me, std::memory_order_relaxed);
    bool share = false;
    if (prev >= 0 && prev != me) {
        share = hipStreamQuery(ctx->slots[(size_t) prev].stream) == hipErrorNotReady;
        (void) hipGetLastError();
    }
    SUBMIT_CHK(launch_exact(s.stream, (u32) ctx->n_cu, db.n_units, ctx->dp, db, ctx->table_g1, s.d_wl, wl_count, wl_count_next, wl_cap, s.res, cap, rawwords, max_seg, share));
    if (timed) SUBMIT_CHK(hipEventRecord(ev[2], s.stream));
    if (compat_g1) {  // the stale rows of this batch's pairs (and of the previous batch's tail) are added again, in file order
        SUBMIT_CHK(launch_g1_apply(s.stream, ctx->table, ctx->g1, db, ctx->dp.min_mer, ctx->g1_carry[ctx->g1_batches & 1], ctx->g1_carry[(ctx->g1_batches + 1) & 1],
                                   ctx->g1_carry_cap));
        ctx->g1_batches++;
    }
#undef SUBMIT_CHK
    s.n_launches++;
    s.n_submits++;
    if (ctx->p.flags & TREW_FLAG_TRACK_PRESSURE)  // 64 bytes into pinned memory; trew_hip_table_pressure reads them there
        HIPCHK(ctx, hipMemcpyAsync(s.h_seen, ctx->table.overflow, kDiagWords * 4, hipMemcpyDeviceToHost, s.stream));
    return 0;
}

extern "C" int trew_hip_submit(trew_hip_ctx *ctx, const trew_hip_batch *batch, int slot) {
    if (!ctx || !batch) return -1;
    if (slot < 0 || slot >= (int) ctx->slots.size()) return fail(ctx, "slot out of range");
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    Slot &s = ctx->slots[(size_t) slot];
    u32 max_seg = 0, max_len = 0;
    if (int rc = batch_geometry(ctx, batch, &max_seg, &max_len)) return rc;
    DevBatch db;
    if (int rc = stage_batch(ctx, batch, s, &db)) return rc;
    return launch_batch(ctx, s, db, max_seg, max_len);
}

// ---- text batches: copy, pack on the device, then the same two kernels
static int stage_ascii(trew_hip_ctx *ctx, const trew_hip_ascii_batch *a, Slot &s, DevBatch *db, u64 *n_words_out) {
    if (!s.d_ascii) return fail(ctx, "the context was created with max_batch_ascii_bytes = 0");
    if (a->n_reads > ctx->p.max_batch_reads) return fail(ctx, "batch has more reads than max_batch_reads");
    if (ctx->p.mode == TREW_MODE_PAIR && (a->n_reads & 1)) return fail(ctx, "pair mode needs an even number of reads");
    const bool ragged = a->word_offsets != nullptr;
    if (ragged != (a->byte_offsets != nullptr) || ragged != (a->lengths != nullptr)) return fail(ctx, "word_offsets, byte_offsets and lengths must all be given or all be NULL");
    u64 n_words = 0;
    if (ragged) {
        // the pack kernel indexes bases[] and words[] with these: validate on the host, as trew_hip_submit does
        u64 w = 0;
        for (u64 i = 0; i < a->n_reads; i++) {
            if (a->word_offsets[i] != w) return fail(ctx, "word_offsets is not the running sum of the packed read sizes");
            if ((u64) a->byte_offsets[i] + a->lengths[i] > a->n_bytes) return fail(ctx, "a read's byte offset / length points outside the batch's bases");
            w += 3ull * (((u64) a->lengths[i] + 31ull) / 32ull);
            if (w > 0xffffffffull) return fail(ctx, "batch packs to more than 2^32 words");
        }
        n_words = w;
        if (a->n_bytes + 12ull * a->n_reads > ctx->p.max_batch_ascii_bytes) return fail(ctx, "text batch larger than max_batch_ascii_bytes");
    } else {
        if (a->uniform_length == 0 && a->n_reads) return fail(ctx, "uniform text batch without uniform_length");
        if (a->n_reads * (u64) a->uniform_length > a->n_bytes) return fail(ctx, "uniform text batch: n_reads * uniform_length exceeds n_bytes");
        if (a->n_bytes > ctx->p.max_batch_ascii_bytes) return fail(ctx, "text batch larger than max_batch_ascii_bytes");
        n_words = a->n_reads * 3ull * (((u64) a->uniform_length + 31ull) / 32ull);
    }
    if (n_words > ctx->p.max_batch_words) return fail(ctx, "batch has more words than max_batch_words");
    *n_words_out = n_words;
    db->n_reads = a->n_reads;
    db->n_units = ctx->p.mode == TREW_MODE_PAIR ? a->n_reads / 2 : a->n_reads;
    db->words = s.d_words;
    if (a->n_reads == 0) return 0;
    const unsigned char *d_bases;
    const u32 *d_wo = nullptr, *d_bo = nullptr, *d_len = nullptr;
    if (ragged) {
        u32 *arr = (u32 *) s.d_ascii;
        d_wo = arr;
        d_bo = arr + a->n_reads;
        d_len = arr + 2 * a->n_reads;
        d_bases = s.d_ascii + 12ull * a->n_reads;
        const bool one = a->byte_offsets == a->word_offsets + a->n_reads && a->lengths == a->byte_offsets + a->n_reads &&
                         (const void *) a->bases == (const void *) (a->lengths + a->n_reads);
        BatchCopy bc(ctx, s);
        if (int rc = bc.begin()) return rc;
        if (one) {  // [word_offsets][byte_offsets][lengths][bases] in one pinned buffer: ONE copy
            if (int rc = bc.copy(s.d_ascii, a->word_offsets, 12ull * a->n_reads + a->n_bytes)) return rc;
        } else {
            if (int rc = bc.copy(arr, a->word_offsets, 4ull * a->n_reads)) return rc;
            if (int rc = bc.copy(arr + a->n_reads, a->byte_offsets, 4ull * a->n_reads)) return rc;
            if (int rc = bc.copy(arr + 2 * a->n_reads, a->lengths, 4ull * a->n_reads)) return rc;
            if (int rc = bc.copy(s.d_ascii + 12ull * a->n_reads, a->bases, a->n_bytes)) return rc;
        }
        if (int rc = bc.end()) return rc;
        db->offsets = d_wo;
        db->lengths = d_len;
        db->uniform_length = 0;
        db->uniform_stride = 0;
    } else {
        d_bases = s.d_ascii;
        BatchCopy bc(ctx, s);
        if (int rc = bc.begin()) return rc;
        if (int rc = bc.copy(s.d_ascii, a->bases, a->n_reads * (u64) a->uniform_length)) return rc;
        if (int rc = bc.end()) return rc;
        db->offsets = nullptr;
        db->lengths = nullptr;
        db->uniform_length = a->uniform_length;
        db->uniform_stride = 3u * ((a->uniform_length + 31u) / 32u);
    }
    HIPCHK(ctx, launch_pack_ascii(s.stream, d_bases, d_bo, d_len, d_wo, a->uniform_length, a->n_reads, n_words / 3ull, s.d_words));
    return 0;
}

static int ascii_geometry(trew_hip_ctx *ctx, const trew_hip_ascii_batch *a, u32 *max_seg, u32 *max_len) {
    trew_hip_batch view;
    memset(&view, 0, sizeof(view));
    view.lengths = a->lengths;
    view.offsets = a->word_offsets;
    view.uniform_length = a->uniform_length;
    view.n_reads = a->n_reads;
    return batch_geometry(ctx, &view, max_seg, max_len);
}

extern "C" int trew_hip_submit_ascii(trew_hip_ctx *ctx, const trew_hip_ascii_batch *batch, int slot) {
    if (!ctx || !batch) return -1;
    if (slot < 0 || slot >= (int) ctx->slots.size()) return fail(ctx, "slot out of range");
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    Slot &s = ctx->slots[(size_t) slot];
    u32 max_seg = 0, max_len = 0;
    if (int rc = ascii_geometry(ctx, batch, &max_seg, &max_len)) return rc;
    DevBatch db;
    u64 n_words = 0;
    if (int rc = stage_ascii(ctx, batch, s, &db, &n_words)) return rc;
    return launch_batch(ctx, s, db, max_seg, max_len);
}

extern "C" int trew_hip_pack_ascii(trew_hip_ctx *ctx, const trew_hip_ascii_batch *batch, uint32_t *words, uint64_t words_cap, uint64_t *n_words) {
    if (!ctx || !batch || !n_words) return -1;
    if (int rc = sync_all(ctx)) return rc;
    Slot &s = ctx->slots[0];
    u32 max_seg = 0, max_len = 0;
    if (int rc = ascii_geometry(ctx, batch, &max_seg, &max_len)) return rc;
    DevBatch db;
    if (int rc = stage_ascii(ctx, batch, s, &db, n_words)) return rc;
    HIPCHK(ctx, hipStreamSynchronize(s.stream));
    if (words && *n_words && *n_words <= words_cap) HIPCHK(ctx, hipMemcpy(words, s.d_words, *n_words * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int trew_hip_wait(trew_hip_ctx *ctx, int slot) {
    if (!ctx) return -1;
    if (slot < 0 || slot >= (int) ctx->slots.size()) return fail(ctx, "slot out of range");
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->slots[(size_t) slot].stream));
    return 0;
}

static int sync_all(trew_hip_ctx *ctx) {
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    for (auto &s : ctx->slots) HIPCHK(ctx, hipStreamSynchronize(s.stream));
    return 0;
}

// reads the device counters; a non-zero "cannot happen" counter is an error, never a silent loss
static int check_diag(trew_hip_ctx *ctx, u32 (&diag)[kDiagWords]) {
    HIPCHK(ctx, hipMemcpy(diag, ctx->table.overflow, sizeof(diag), hipMemcpyDeviceToHost));
    note_seen(ctx, diag);
    if (diag[kDiagOverflow]) return fail(ctx, "device count table and its spill log are full: raise table_log2_slots");
    if (diag[kDiagWorklistDrop]) return fail(ctx, "internal error: the prefilter worklist overflowed (survivors were dropped)");
    if (diag[kDiagIntentDrop]) return fail(ctx, "internal error: a pair logged more than 32 deferred emissions (some were dropped)");
    if (diag[kDiagG1Drop]) return fail(ctx, "TREW_FLAG_COMPAT_G1: the stale-row log of a batch overflowed (submit smaller batches)");
    if (diag[kDiagKernarg]) return fail(ctx, "internal error: the exact kernel's table descriptor differs from its by-value argument (kernarg layout)");
    return 0;
}

static int ensure_rows(trew_hip_ctx *ctx, trew_hip_row **buf, u64 *cap, u64 want) {
    if (want <= *cap) return 0;
    if (*buf) (void) hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    want = std::max<u64>(want + want / 4, 1ull << 16);
    HIPCHK(ctx, hipMalloc((void **) buf, want * sizeof(trew_hip_row)));
    *cap = want;
    return 0;
}

// Compacts the occupied slots of `table` (-1: all) followed by the spill log into d_rows (device memory, capacity
// cap rows) and reports the number of rows there are.  Rows may repeat a key (spilled rows, wide duplicates).
// Caller holds table_mu and has synchronised the slots.
static int compact_to(trew_hip_ctx *ctx, int table, trew_hip_row *d_rows, u64 cap, u64 *n_rows) {
    u32 diag[kDiagWords];
    if (int rc = check_diag(ctx, diag)) return rc;
    u32 n_spill = 0;
    HIPCHK(ctx, hipMemcpy(&n_spill, ctx->wide.spill_n, 4, hipMemcpyDeviceToHost));
    n_spill = std::min(n_spill, ctx->wide.spill_cap);
    if (!ctx->d_collect_n) HIPCHK(ctx, hipMalloc((void **) &ctx->d_collect_n, 8));
    hipStream_t st = ctx->slots[0].stream;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_collect_n, 0, 8, st));
    HIPCHK(ctx, launch_compact(st, ctx->table, ctx->table_slots, ctx->wide.wide_log2_slots, table, d_rows, d_rows ? cap : 0, ctx->d_collect_n));
    unsigned long long n = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n, ctx->d_collect_n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (n_spill) {
        if (table < 0) {  // the log is device-resident: append it as it is
            if (d_rows && n + n_spill <= cap)
                {  // on the stream the compaction ran on, and complete before anyone reads d_rows (callers use other streams)
                    HIPCHK(ctx, hipMemcpyAsync(d_rows + n, ctx->wide.spill_rows, (size_t) n_spill * sizeof(trew_hip_row), hipMemcpyDeviceToDevice, st));
                    HIPCHK(ctx, hipStreamSynchronize(st));
                }
            n += n_spill;
        } else {  // one table only: select on the host (rare path)
            std::vector<trew_hip_row> sp(n_spill);
            HIPCHK(ctx, hipMemcpy(sp.data(), ctx->wide.spill_rows, (size_t) n_spill * sizeof(trew_hip_row), hipMemcpyDeviceToHost));
            for (const auto &r : sp) {
                if (r.table != table) continue;
                if (d_rows && n < cap) HIPCHK(ctx, hipMemcpy(d_rows + n, &r, sizeof(r), hipMemcpyHostToDevice));
                n++;
            }
        }
    }
    *n_rows = n;
    return 0;
}

extern "C" int trew_hip_collect(trew_hip_ctx *ctx, int table, trew_hip_row *rows, uint64_t cap, uint64_t *n_rows) {
    if (!ctx || !n_rows) return -1;
    if (table < -1 || table >= TREW_NUM_TABLES) return fail(ctx, "table out of range");
    std::lock_guard<std::mutex> lk(ctx->table_mu);
    if (int rc = sync_all(ctx)) return rc;
    // compact on the device, copy only the occupied rows (scratch buffers persist across calls)
    const u64 dcap = rows ? cap : 0;
    if (int rc = ensure_rows(ctx, &ctx->d_collect_rows, &ctx->collect_cap, dcap)) return rc;
    u64 n = 0;
    if (int rc = compact_to(ctx, table, dcap ? ctx->d_collect_rows : nullptr, dcap, &n)) return rc;
    if (dcap && n && n <= dcap) {
        HIPCHK(ctx, hipMemcpy(rows, ctx->d_collect_rows, n * sizeof(trew_hip_row), hipMemcpyDeviceToHost));
        // the wide-entry protocol may leave one key in two slots (see table_add_wide), spilled rows repeat
        // keys: counts are sums, merge them
        u32 n_spill = 0;
        HIPCHK(ctx, hipMemcpy(&n_spill, ctx->wide.spill_n, 4, hipMemcpyDeviceToHost));
        bool any_dup = n_spill != 0;
        for (u64 i = 0; i < n && !any_dup; i++) any_dup = rows[i].k > 32;
        if (any_dup) {
            std::sort(rows, rows + n, [](const trew_hip_row &a, const trew_hip_row &b) {
                if (a.table != b.table) return a.table < b.table;
                if (a.k != b.k) return a.k < b.k;
                if (a.word_hi != b.word_hi) return a.word_hi < b.word_hi;
                return a.word_lo < b.word_lo;
            });
            u64 m = 0;
            for (u64 i = 0; i < n; i++) {
                if (m && rows[m - 1].table == rows[i].table && rows[m - 1].k == rows[i].k && rows[m - 1].word_hi == rows[i].word_hi &&
                    rows[m - 1].word_lo == rows[i].word_lo)
                    rows[m - 1].count += rows[i].count;
                else
                    rows[m++] = rows[i];
            }
            // a key whose counts were moved elsewhere (run_long subtracts what it re-files) may sum to zero: not a row
            u64 z = 0;
            for (u64 i = 0; i < m; i++)
                if (rows[i].count) rows[z++] = rows[i];
            n = z;
        }
    }
    *n_rows = n;
    return 0;
}

extern "C" int trew_hip_collect_device(trew_hip_ctx *ctx, trew_hip_row *d_rows, uint64_t cap, uint64_t *n_rows) {
    if (!ctx || !n_rows) return -1;
    std::lock_guard<std::mutex> lk(ctx->table_mu);
    if (int rc = sync_all(ctx)) return rc;
    HIPCHK(ctx, hipDeviceSynchronize());  // the caller's buffer may still be in use on a stream of its own (torch)
    return compact_to(ctx, -1, d_rows, d_rows ? cap : 0, n_rows);
}

extern "C" int trew_hip_collect_slice_device(trew_hip_ctx *ctx, trew_hip_row *d_slice, uint64_t slice_rows, void *consumer_stream, uint64_t *n_rows) {
    if (!ctx || !d_slice) return -1;
    std::lock_guard<std::mutex> lk(ctx->table_mu);
    if (int rc = sync_all(ctx)) return rc;
    hipStream_t st = ctx->slots[0].stream;
    if (consumer_stream) {  // whatever the caller's stream still does with the slice (a previous collective) comes first
        HIPCHK(ctx, hipEventRecord(ctx->ev_producer, (hipStream_t) consumer_stream));
        HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_producer, 0));
    } else {
        HIPCHK(ctx, hipDeviceSynchronize());
    }
    if (!ctx->d_collect_n) HIPCHK(ctx, hipMalloc((void **) &ctx->d_collect_n, 8));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_collect_n, 0, 8, st));
    HIPCHK(ctx, launch_compact(st, ctx->table, ctx->table_slots, ctx->wide.wide_log2_slots, -1, d_slice + 1, slice_rows, ctx->d_collect_n));
    HIPCHK(ctx, launch_slice_finish(st, d_slice, slice_rows, ctx->d_collect_n, ctx->wide.spill_rows, ctx->wide.spill_n, ctx->wide.spill_cap));
    if (consumer_stream) {  // the collective is ordered behind the header on the device: the host is not in the way
        HIPCHK(ctx, hipEventRecord(ctx->ev_producer, st));
        HIPCHK(ctx, hipStreamWaitEvent((hipStream_t) consumer_stream, ctx->ev_producer, 0));
    }
    if (n_rows || !consumer_stream) {
        trew_hip_row h;
        memset(&h, 0, sizeof(h));
        HIPCHK(ctx, hipMemcpyAsync(&h, d_slice, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (n_rows) *n_rows = h.count;
    }
    return 0;
}

static int reset_locked(trew_hip_ctx *ctx) {
    // the callers have synchronised every slot, so nothing reads or writes the tables; the fills are queued on the fill stream
    // and every stream of the context waits for them on the device (see fill_stream) -- no host wait
    hipStream_t fs = ctx->fill_stream;
    HIPCHK(ctx, hipMemsetAsync(ctx->table.keys, 0, ctx->table_slots * 8, fs));
    HIPCHK(ctx, hipMemsetAsync(ctx->table.counts, 0, ctx->table_slots * 8, fs));
    HIPCHK(ctx, hipMemsetAsync(ctx->table.overflow, 0, kDiagWords * 4, fs));
    const size_t wb = (size_t) 8 << ctx->wide.wide_log2_slots;
    HIPCHK(ctx, hipMemsetAsync(ctx->wide.wtag, 0, wb, fs));
    HIPCHK(ctx, hipMemsetAsync(ctx->wide.wlo, 0, wb, fs));
    HIPCHK(ctx, hipMemsetAsync(ctx->wide.whi, 0, wb, fs));
    HIPCHK(ctx, hipMemsetAsync(ctx->wide.wcount, 0, wb, fs));
    HIPCHK(ctx, fallback_counters_clear(fs));
    if (ctx->g1.counters) HIPCHK(ctx, hipMemsetAsync(ctx->g1.counters, 0, 16, fs));  // a new input: nothing is left in the stale map
    if (int rc = order_behind_fills(ctx)) return rc;
    // the callers have synchronised every slot: no copy into h_seen is in flight
    std::lock_guard<std::mutex> lk(ctx->seen_mu);
    for (auto &sl : ctx->slots) memset(sl.h_seen, 0, kDiagWords * 4);
    for (auto &v : ctx->seen) v.store(0, std::memory_order_relaxed);
    return 0;
}

extern "C" int trew_hip_reset_tables(trew_hip_ctx *ctx) {
    if (!ctx) return -1;
    std::lock_guard<std::mutex> lk(ctx->table_mu);
    if (int rc = sync_all(ctx)) return rc;
    return reset_locked(ctx);
}

extern "C" int trew_hip_table_pressure(trew_hip_ctx *ctx, uint64_t *used_slots, uint64_t *total_slots, uint64_t *spilled_rows,
                                       uint64_t *spill_capacity) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    u32 diag[kDiagWords];
    if (ctx->p.flags & TREW_FLAG_TRACK_PRESSURE) {
        // every slot's copy, finished or still being written: the counters only grow between resets and a copy lands a
        // whole word at a time, so any mix of old and new words is a valid (slightly old) reading
        std::lock_guard<std::mutex> lk(ctx->seen_mu);
        for (auto &sl : ctx->slots) {
            u32 copy[kDiagWords];
            for (int i = 0; i < kDiagWords; i++) copy[i] = ((volatile const u32 *) sl.h_seen)[i];
            note_seen(ctx, copy);
        }
        for (int i = 0; i < kDiagWords; i++) diag[i] = ctx->seen[i].load(std::memory_order_relaxed);
    } else {
        // a plain blocking copy: a snapshot of monotonic counters, the slot streams are not waited for (a reset still in flight is)
        HIPCHK(ctx, hipStreamSynchronize(ctx->fill_stream));
        HIPCHK(ctx, hipMemcpy(diag, ctx->table.overflow, sizeof(diag), hipMemcpyDeviceToHost));
    }
    const u32 n_spill = diag[kDiagSpillRows];
    // the narrow and the wide table fill independently: report the fuller one, scaled to the narrow table's size
    const u64 wide_slots = 1ull << ctx->wide.wide_log2_slots;
    const u64 wide_scaled = (u64) ((double) diag[kDiagInsertedWide] / (double) wide_slots * (double) ctx->table_slots);
    if (used_slots) *used_slots = std::max<u64>(diag[kDiagInserted], wide_scaled);
    if (total_slots) *total_slots = ctx->table_slots;
    if (spilled_rows) *spilled_rows = diag[kDiagOverflow] ? (u64) ctx->wide.spill_cap + 1 : n_spill;
    if (spill_capacity) *spill_capacity = ctx->wide.spill_cap;
    return 0;
}

// Adds n_rows device-resident rows on slot 0's stream.  validate: run the device-side check first (rows of unknown origin);
// the add is all or nothing -- with a bad row nothing is added and the call fails, the context stays usable.
static int add_device_rows_locked(trew_hip_ctx *ctx, const trew_hip_row *d_rows, u64 n_rows, bool validate) {
    hipStream_t st = ctx->slots[0].stream;
    if (validate) HIPCHK(ctx, hipMemsetAsync(ctx->d_row_flags, 0, kRowFlagWords * 4, st));
    HIPCHK(ctx, launch_add_rows(st, ctx->table, d_rows, n_rows, validate ? ctx->d_row_flags : nullptr));
    u32 flags[kRowFlagWords] = {0, 0, 0, 0};
    if (validate) HIPCHK(ctx, hipMemcpyAsync(flags, ctx->d_row_flags, sizeof(flags), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (flags[kRowFlagBad]) return fail(ctx, "trew_hip_add_rows_device: row out of range (nothing was added)");
    u32 diag[kDiagWords];
    return check_diag(ctx, diag);
}

extern "C" int trew_hip_add_rows(trew_hip_ctx *ctx, const trew_hip_row *rows, uint64_t n_rows) {
    if (!ctx) return -1;
    if (n_rows == 0) return 0;
    for (u64 i = 0; i < n_rows; i++)
        if (rows[i].k < 1 || rows[i].k > 64 || rows[i].table < 0 || rows[i].table >= TREW_NUM_TABLES || (rows[i].k <= 32 && rows[i].word_hi))
            return fail(ctx, "trew_hip_add_rows: row out of range");
    std::lock_guard<std::mutex> lk(ctx->table_mu);
    if (int rc = sync_all(ctx)) return rc;
    if (int rc = ensure_rows(ctx, &ctx->d_add_rows, &ctx->add_cap, n_rows)) return rc;  // persistent scratch: no malloc/free per call
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_add_rows, rows, n_rows * sizeof(trew_hip_row), hipMemcpyHostToDevice, ctx->slots[0].stream));
    return add_device_rows_locked(ctx, ctx->d_add_rows, n_rows, false);  // validated above, on the host
}

extern "C" int trew_hip_add_rows_device(trew_hip_ctx *ctx, const trew_hip_row *d_rows, uint64_t n_rows) {
    if (!ctx) return -1;
    if (n_rows == 0) return 0;
    if (!d_rows) return fail(ctx, "trew_hip_add_rows_device: null rows");
    std::lock_guard<std::mutex> lk(ctx->table_mu);
    if (int rc = sync_all(ctx)) return rc;
    HIPCHK(ctx, hipDeviceSynchronize());  // the rows may have been produced on a stream of the caller's (an RCCL all_gather)
    return add_device_rows_locked(ctx, d_rows, n_rows, true);
}

extern "C" int trew_hip_add_gathered_device(trew_hip_ctx *ctx, const trew_hip_row *d_buf, uint32_t n_slices, uint32_t own_slice,
                                            uint64_t slice_rows, void *producer_stream, uint64_t *max_rows) {
    if (!ctx) return -1;
    if (!d_buf || n_slices == 0 || own_slice >= n_slices) return fail(ctx, "trew_hip_add_gathered_device: bad arguments");
    std::lock_guard<std::mutex> lk(ctx->table_mu);
    if (int rc = sync_all(ctx)) return rc;
    hipStream_t st = ctx->slots[0].stream;
    if (producer_stream) {  // device-side ordering behind the collective: no host synchronisation between the two
        HIPCHK(ctx, hipEventRecord(ctx->ev_producer, (hipStream_t) producer_stream));
        HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_producer, 0));
    } else {
        HIPCHK(ctx, hipDeviceSynchronize());
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_row_flags, 0, kRowFlagWords * 4, st));
    HIPCHK(ctx, launch_add_gathered(st, ctx->table, d_buf, n_slices, own_slice, slice_rows, ctx->d_row_flags));
    u32 flags[kRowFlagWords] = {0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(flags, ctx->d_row_flags, sizeof(flags), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (max_rows) *max_rows = ((u64) flags[kRowFlagMaxHi] << 32) | flags[kRowFlagMaxLo];
    if (flags[kRowFlagBad]) return fail(ctx, "trew_hip_add_gathered_device: row out of range (nothing was added)");
    if (flags[kRowFlagOverflow]) return 0;  // a slice holds more rows than fit: nothing was added, *max_rows says how many there are
    u32 diag[kDiagWords];
    return check_diag(ctx, diag);
}

extern "C" int trew_hip_debug_worklist(trew_hip_ctx *ctx, int slot, uint32_t *units, uint64_t cap, uint64_t *n) {
    if (!ctx || !n) return -1;
    if (slot < 0 || slot >= (int) ctx->slots.size()) return fail(ctx, "slot out of range");
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    Slot &s = ctx->slots[(size_t) slot];
    HIPCHK(ctx, hipStreamSynchronize(s.stream));
    u32 c = 0;
    // the counter block of the last launch keeps its worklist size: the NEXT launch's exact kernel clears it (see exact_kernel)
    if (s.n_launches) HIPCHK(ctx, hipMemcpy(&c, s.d_wl_count + ((s.n_launches - 1) & 1) * kWlCountWords, 4, hipMemcpyDeviceToHost));
    c = (u32) std::min<u64>(c, ctx->p.max_batch_reads);
    *n = c;
    const u64 take = std::min<u64>(c, cap);
    if (units && take) HIPCHK(ctx, hipMemcpy(units, s.d_wl, take * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int trew_hip_debug_counters(trew_hip_ctx *ctx, uint64_t *out, int n) {
    if (!ctx || !out || n < 0) return -1;
    if (int rc = sync_all(ctx)) return rc;
    u32 diag[kDiagWords], fb[kFallbackWords];
    HIPCHK(ctx, hipMemcpy(diag, ctx->table.overflow, sizeof(diag), hipMemcpyDeviceToHost));
    HIPCHK(ctx, fallback_counters_read(fb));
    const u32 v[TREW_DEBUG_COUNTERS] = {fb[kFallbackStrictRerun], fb[kFallbackWindows], fb[kFallbackWideSpin], diag[kDiagInserted], diag[kDiagInsertedWide], fb[kFallbackGroupPunt], fb[kFallbackGroupRouted], fb[kFallbackGroupTarget]};
    for (int i = 0; i < n; i++) out[i] = i < TREW_DEBUG_COUNTERS ? v[i] : 0;
    return 0;
}

extern "C" int trew_hip_merge(trew_hip_ctx *dst, trew_hip_ctx *src) {
    if (!dst || !src) return -1;
    if (dst == src) return fail(dst, "trew_hip_merge: source and destination are the same context");
    // lock order by address: two threads merging in opposite directions cannot deadlock
    std::mutex *m1 = &dst->table_mu, *m2 = &src->table_mu;
    if (m2 < m1) std::swap(m1, m2);
    std::lock_guard<std::mutex> l1(*m1), l2(*m2);
    if (int rc = sync_all(src)) return rc;
    u64 n = 0;
    if (int rc = compact_to(src, -1, nullptr, 0, &n)) return rc;  // size first
    if (n == 0) return 0;
    if (int rc = ensure_rows(src, &src->d_collect_rows, &src->collect_cap, n)) return rc;
    if (int rc = compact_to(src, -1, src->d_collect_rows, src->collect_cap, &n)) return rc;
    if (int rc = sync_all(dst)) return rc;
    if (int rc = ensure_rows(dst, &dst->d_add_rows, &dst->add_cap, n)) return rc;
    // device to device: over xGMI when the contexts sit on two GPUs, a plain copy when they share one
    // (queued on the stream that launches the add kernel, so the kernel is ordered behind the copy by the stream itself;
    // src's rows are complete: compact_to synchronised its stream)
    HIPCHK(dst, hipSetDevice(dst->p.device));
    HIPCHK(dst, hipMemcpyPeerAsync(dst->d_add_rows, dst->p.device, src->d_collect_rows, src->p.device, n * sizeof(trew_hip_row), dst->slots[0].stream));
    return add_device_rows_locked(dst, dst->d_add_rows, n, false);  // rows compacted from a table of this library
}

extern "C" int trew_hip_segment_results(trew_hip_ctx *ctx, int slot, int32_t *k_high, int32_t *k_low,
                                        uint64_t *seq_high, uint64_t *seq_low, uint64_t *seq_high_hi, uint64_t *seq_low_hi,
                                        uint64_t n_reads) {
    if (!ctx) return -1;
    if (ctx->p.mode != TREW_MODE_SEGMENT) return fail(ctx, "segment results exist only in TREW_MODE_SEGMENT");
    if (slot < 0 || slot >= (int) ctx->slots.size()) return fail(ctx, "slot out of range");
    if (int rc = trew_hip_wait(ctx, slot)) return rc;
    Slot &s = ctx->slots[(size_t) slot];
    if (n_reads > s.n_units) return fail(ctx, "n_reads exceeds the last batch");
    if (k_high) HIPCHK(ctx, hipMemcpy(k_high, s.res.k_high, n_reads * 4, hipMemcpyDeviceToHost));
    if (k_low) HIPCHK(ctx, hipMemcpy(k_low, s.res.k_low, n_reads * 4, hipMemcpyDeviceToHost));
    if (seq_high) HIPCHK(ctx, hipMemcpy(seq_high, s.res.seq_high, n_reads * 8, hipMemcpyDeviceToHost));
    if (seq_low) HIPCHK(ctx, hipMemcpy(seq_low, s.res.seq_low, n_reads * 8, hipMemcpyDeviceToHost));
    if (seq_high_hi) HIPCHK(ctx, hipMemcpy(seq_high_hi, s.res.seq_high_hi, n_reads * 8, hipMemcpyDeviceToHost));
    if (seq_low_hi) HIPCHK(ctx, hipMemcpy(seq_low_hi, s.res.seq_low_hi, n_reads * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int trew_hip_filter_masks(trew_hip_ctx *ctx, const trew_hip_batch *batch, uint64_t *cand, int slots_per_read) {
    if (!ctx || !batch || !cand) return -1;
    if (slots_per_read < 1 || slots_per_read > kMaxSlots) return fail(ctx, "slots_per_read out of range");
    if (int rc = sync_all(ctx)) return rc;
    Slot &s = ctx->slots[0];
    u32 max_seg = 0, max_len = 0;
    if (int rc = batch_geometry(ctx, batch, &max_seg, &max_len)) return rc;
    DevBatch db;
    if (int rc = stage_batch(ctx, batch, s, &db)) return rc;
    if (db.n_units == 0) return 0;
    u64 *d = nullptr;
    const u64 bytes = db.n_units * (u64) slots_per_read * 8ull;
    HIPCHK(ctx, hipMalloc((void **) &d, bytes));
    u32 *const wl_count = s.d_wl_count + (s.n_launches & 1) * kWlCountWords;
    const int2 *d_thr = nullptr;
    if (int rc = stage_thresholds(ctx, s, db, &d_thr)) {
        (void) hipFree(d);
        return rc;
    }
    hipError_t e = hipMemsetAsync(d, 0, bytes, s.stream);
    if (e == hipSuccess)
        e = launch_filter(s.stream, (u32) ctx->n_cu, max_seg, ctx->dp, db, s.d_wl, wl_count, (u32) ctx->p.max_batch_reads, d, slots_per_read,
                          ctx->table.overflow, d_thr);
    if (e == hipSuccess) e = hipMemsetAsync(wl_count, 0, kWlCountBytes, s.stream);  // no exact kernel follows: leave the block clean
    if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
    if (e == hipSuccess) e = hipMemcpy(cand, d, bytes, hipMemcpyDeviceToHost);
    (void) hipFree(d);
    HIPCHK(ctx, e);
    return 0;
}

extern "C" int trew_hip_last_timing(trew_hip_ctx *ctx, int slot, float *ms_filter, float *ms_exact, uint64_t *n_flagged) {
    if (!ctx) return -1;
    if (slot < 0 || slot >= (int) ctx->slots.size()) return fail(ctx, "slot out of range");
    if (ctx->p.flags & TREW_FLAG_NO_TIMING) return fail(ctx, "the context was created with TREW_FLAG_NO_TIMING");
    if (int rc = trew_hip_wait(ctx, slot)) return rc;
    Slot &s = ctx->slots[(size_t) slot];
    // mean over the submits since the previous call (at most the last kRing of them)
    u64 first = s.n_reported;
    if (s.n_submits - first > (u64) Slot::kRing) first = s.n_submits - Slot::kRing;
    if (first == s.n_submits) return fail(ctx, "no timed submit on this slot since the last query");
    double sf = 0, se = 0;
    for (u64 i = first; i < s.n_submits; i++) {
        float a = 0, b = 0;
        hipEvent_t *ev = s.ev[i % Slot::kRing];
        HIPCHK(ctx, hipEventElapsedTime(&a, ev[0], ev[1]));
        HIPCHK(ctx, hipEventElapsedTime(&b, ev[1], ev[2]));
        sf += a;
        se += b;
    }
    const double cnt = (double) (s.n_submits - first);
    s.n_reported = s.n_submits;
    if (ms_filter) *ms_filter = (float) (sf / cnt);
    if (ms_exact) *ms_exact = (float) (se / cnt);
    if (n_flagged) {  // optional: costs one blocking 4-byte copy
        u32 c = 0;
        if (s.n_launches) HIPCHK(ctx, hipMemcpy(&c, s.d_wl_count + ((s.n_launches - 1) & 1) * kWlCountWords, 4, hipMemcpyDeviceToHost));
        *n_flagged = c;
    }
    return 0;
}

// ---------------------------------------------------------------- host packing
extern "C" uint64_t trew_pack_words(uint64_t n_bases) { return 3ull * ((n_bases + 31ull) / 32ull); }

namespace {
struct PackLut {
    unsigned char v[256];
    PackLut() {
        // codes[], kmer.cpp:14-31: T=0 G=1 C=2 A=3, either case; everything else is "N"
        for (int i = 0; i < 256; i++) v[i] = 4;
        v[(int) 'T'] = v[(int) 't'] = 0;
        v[(int) 'G'] = v[(int) 'g'] = 1;
        v[(int) 'C'] = v[(int) 'c'] = 2;
        v[(int) 'A'] = v[(int) 'a'] = 3;
    }
};
const PackLut g_lut;

// 32 bases -> {lo, hi, nmask} with AVX2 compares + movemask (about 10x the table loop)
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) inline void pack32_avx2(const unsigned char *q, uint32_t *out) {
    const __m256i c = _mm256_and_si256(_mm256_loadu_si256((const __m256i *) q), _mm256_set1_epi8((char) 0xDF));  // upper-case
    const __m256i a = _mm256_cmpeq_epi8(c, _mm256_set1_epi8('A')), cc = _mm256_cmpeq_epi8(c, _mm256_set1_epi8('C'));
    const __m256i g = _mm256_cmpeq_epi8(c, _mm256_set1_epi8('G')), t = _mm256_cmpeq_epi8(c, _mm256_set1_epi8('T'));
    // letters only: 0xDF folding must not turn e.g. 'a'-0x20 twins of non-letters into hits: A/C/G/T & a/c/g/t are the only preimages
    const uint32_t ma = (uint32_t) _mm256_movemask_epi8(a), mc = (uint32_t) _mm256_movemask_epi8(cc);
    const uint32_t mg = (uint32_t) _mm256_movemask_epi8(g), mt = (uint32_t) _mm256_movemask_epi8(t);
    out[0] = mg | ma;             // lo bit: G=1, A=3
    out[1] = mc | ma;             // hi bit: C=2, A=3
    out[2] = ~(ma | mc | mg | mt);  // everything else is "N"
}
// 64 bases -> two triples with AVX-512BW byte compares straight into mask registers
__attribute__((target("avx512f,avx512bw"))) inline void pack64_avx512(const unsigned char *q, uint32_t *out) {
    const __m512i c = _mm512_and_si512(_mm512_loadu_si512((const void *) q), _mm512_set1_epi8((char) 0xDF));
    const uint64_t ma = _mm512_cmpeq_epi8_mask(c, _mm512_set1_epi8('A')), mc = _mm512_cmpeq_epi8_mask(c, _mm512_set1_epi8('C'));
    const uint64_t mg = _mm512_cmpeq_epi8_mask(c, _mm512_set1_epi8('G')), mt = _mm512_cmpeq_epi8_mask(c, _mm512_set1_epi8('T'));
    const uint64_t lo = mg | ma, hi = mc | ma, nm = ~(ma | mc | mg | mt);
    out[0] = (uint32_t) lo;
    out[1] = (uint32_t) hi;
    out[2] = (uint32_t) nm;
    out[3] = (uint32_t) (lo >> 32);
    out[4] = (uint32_t) (hi >> 32);
    out[5] = (uint32_t) (nm >> 32);
}
#endif

// one read -> triples; returns words written
inline uint64_t pack_one(const unsigned char *p, uint64_t len, uint32_t *out) {
    const uint64_t nw = (len + 31) / 32;
    uint64_t j = 0;
#if defined(__x86_64__)
    static const bool have_avx512 = __builtin_cpu_supports("avx512bw") && !getenv("TREW_NO_AVX512");
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx512)
        for (; 32 * (j + 2) <= len; j += 2) pack64_avx512(p + 32 * j, out + 3 * j);
    if (have_avx2)
        for (; 32 * (j + 1) <= len; j++) pack32_avx2(p + 32 * j, out + 3 * j);
#endif
    for (; j < nw; j++) {
        uint32_t lo = 0, hi = 0, nm = 0;
        const uint64_t m = std::min<uint64_t>(32, len - 32 * j);
        const unsigned char *q = p + 32 * j;
        for (uint64_t i = 0; i < m; i++) {
            const unsigned c = g_lut.v[q[i]];
            lo |= (uint32_t) (c & 1u) << i;
            hi |= (uint32_t) ((c >> 1) & 1u) << i;
            nm |= (uint32_t) (c >> 2) << i;
        }
        out[3 * j + 0] = lo & ~nm;
        out[3 * j + 1] = hi & ~nm;
        out[3 * j + 2] = nm;
    }
    return 3 * nw;
}
}  // namespace

extern "C" uint64_t trew_pack_reads(const char *buf, const int64_t *st, const int64_t *nd, uint64_t n_reads,
                                    uint32_t *words, uint64_t words_cap, uint32_t *offsets, uint32_t *lengths) {
    uint64_t w = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const int64_t n = nd[r] - st[r] + 1;
        const uint64_t len = n > 0 ? (uint64_t) n : 0;
        if (w + 3 * ((len + 31) / 32) > words_cap || w > 0xffffffffull) return (uint64_t) -1;
        offsets[r] = (uint32_t) w;
        lengths[r] = (uint32_t) len;
        w += pack_one((const unsigned char *) buf + st[r], len, words + w);
    }
    return w;
}

extern "C" uint64_t trew_pack_pairs(const char *buf1, const int64_t *st1, const int64_t *nd1,
                                    const char *buf2, const int64_t *st2, const int64_t *nd2, uint64_t n_pairs,
                                    uint32_t *words, uint64_t words_cap, uint32_t *offsets, uint32_t *lengths) {
    uint64_t w = 0;
    for (uint64_t r = 0; r < n_pairs; r++) {
        for (int mate = 0; mate < 2; mate++) {
            const char *buf = mate ? buf2 : buf1;
            const int64_t s = mate ? st2[r] : st1[r], e = mate ? nd2[r] : nd1[r];
            const uint64_t len = e - s + 1 > 0 ? (uint64_t) (e - s + 1) : 0;
            if (w + 3 * ((len + 31) / 32) > words_cap || w > 0xffffffffull) return (uint64_t) -1;
            offsets[2 * r + mate] = (uint32_t) w;
            lengths[2 * r + mate] = (uint32_t) len;
            w += pack_one((const unsigned char *) buf + s, len, words + w);
        }
    }
    return w;
}

// ---------------------------------------------------------------- synthetic workloads
template <typename F>
static void parallel_reads(uint64_t n, F fn) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (n < 4096 || nt == 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    const uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t lo = std::min<uint64_t>(n, per * t), hi = std::min<uint64_t>(n, lo + per);
        if (lo < hi) th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto &x : th) x.join();
}

extern "C" int trew_synth_short_ascii(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, char *out) {
    parallel_reads(n_reads, [=](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; r++) {
            const trew_synth::ReadClass c = trew_synth::read_class(seed, first_read + r);
            char *o = out + r * (uint64_t) (read_len + 1);
            for (uint32_t p = 0; p < read_len; p++)
                o[p] = trew_synth::base_char(trew_synth::short_base(seed, first_read + r, c, p, read_len));
            o[read_len] = '\n';
        }
    });
    return 0;
}

extern "C" int trew_synth_pair_ascii(uint64_t seed, uint64_t first_pair, uint64_t n_pairs, uint32_t read_len, char *out1, char *out2) {
    parallel_reads(n_pairs, [=](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; r++) {
            const trew_synth::ReadClass c = trew_synth::read_class(seed, first_pair + r);
            char *o1 = out1 + r * (uint64_t) (read_len + 1);
            char *o2 = out2 + r * (uint64_t) (read_len + 1);
            for (uint32_t p = 0; p < read_len; p++) {
                o1[p] = trew_synth::base_char(trew_synth::pair_base(seed, first_pair + r, c, 0, p, read_len));
                o2[p] = trew_synth::base_char(trew_synth::pair_base(seed, first_pair + r, c, 1, p, read_len));
            }
            o1[read_len] = '\n';
            o2[read_len] = '\n';
        }
    });
    return 0;
}

// ---- long reads: quantile table of clip(lognormal(9.413, 0.7), 1000, 200000), host double math only
static double inv_norm_cdf(double p) {  // Acklam's rational approximation (|rel err| < 1.2e-9)
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    const double pl = 0.02425;
    if (p < pl) {
        const double q = std::sqrt(-2 * std::log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    if (p > 1 - pl) return -inv_norm_cdf(1 - p);
    const double q = p - 0.5, r = q * q;
    return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
}

static const uint32_t *long_quantiles() {
    static uint32_t table[trew_synth::kLongQuantiles];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < trew_synth::kLongQuantiles; i++) {
            const double z = inv_norm_cdf((i + 0.5) / trew_synth::kLongQuantiles);
            double len = std::exp(9.413 + 0.7 * z);
            len = std::min(200000.0, std::max(1000.0, len));
            table[i] = (uint32_t) len;
        }
        init = true;
    }
    return table;
}

extern "C" int trew_synth_long_lengths(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t *lengths) {
    const uint32_t *qt = long_quantiles();
    for (uint64_t r = 0; r < n_reads; r++) lengths[r] = trew_synth::long_class(seed, first_read + r, qt).len;
    return 0;
}

extern "C" int trew_synth_long_ascii(uint64_t seed, uint64_t first_read, uint64_t n_reads, const uint64_t *byte_offsets, char *out) {
    const uint32_t *qt = long_quantiles();
    parallel_reads(n_reads, [=](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; r++) {
            const trew_synth::LongClass c = trew_synth::long_class(seed, first_read + r, qt);
            char *o = out + byte_offsets[r];
            for (uint32_t p = 0; p < c.len; p++) o[p] = trew_synth::base_char(trew_synth::long_base(seed, first_read + r, c, p));
            o[c.len] = '\n';
        }
    });
    return 0;
}

extern "C" int trew_synth_long_device(trew_hip_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                      const uint32_t *d_offsets, uint32_t *d_words) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    uint32_t *d_qt = nullptr;
    HIPCHK(ctx, hipMalloc((void **) &d_qt, trew_synth::kLongQuantiles * 4));
    hipError_t e = hipMemcpy(d_qt, long_quantiles(), trew_synth::kLongQuantiles * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_synth_long(ctx->slots[0].stream, seed, first_read, n_reads, d_qt, d_offsets, d_words);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->slots[0].stream);
    (void) hipFree(d_qt);
    HIPCHK(ctx, e);
    return 0;
}

extern "C" int trew_synth_short_device(trew_hip_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                       uint32_t read_len, uint32_t *d_words) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    HIPCHK(ctx, launch_synth_short(ctx->slots[0].stream, seed, first_read, n_reads, read_len, d_words));
    HIPCHK(ctx, hipStreamSynchronize(ctx->slots[0].stream));
    return 0;
}

extern "C" int trew_synth_pair_device(trew_hip_ctx *ctx, uint64_t seed, uint64_t first_pair, uint64_t n_pairs,
                                      uint32_t read_len, uint32_t *d_words) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    HIPCHK(ctx, launch_synth_pair(ctx->slots[0].stream, seed, first_pair, n_pairs, read_len, d_words));
    HIPCHK(ctx, hipStreamSynchronize(ctx->slots[0].stream));
    return 0;
}

// ---------------------------------------------------------------- device memory helpers
extern "C" int trew_hip_malloc(trew_hip_ctx *ctx, uint64_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    HIPCHK(ctx, hipMalloc(d_ptr, bytes));
    return 0;
}
extern "C" int trew_hip_free(trew_hip_ctx *ctx, void *d_ptr) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    HIPCHK(ctx, hipFree(d_ptr));
    return 0;
}
extern "C" int trew_hip_memcpy_h2d(trew_hip_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    HIPCHK(ctx, hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int trew_hip_memcpy_d2h(trew_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->p.device));
    HIPCHK(ctx, hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
