// trew_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the TREW tandem-repeat scan.
//
// Two kernels per batch, both pure integer (no MFMA -- this is bit/byte work):
//
//  1. filter_kernel<NW>  one LANE per read (persistent blocks).  Bit-parallel upper
//     bound on max-class/COUNT for every k in [MIN_MER, MAX_MER] of every
//     segment of the read.  Windows in one rotation class (get_rot_seq,
//     kmer.cpp:1815-1823) have the same base composition, hence the same three
//     parities (#lo-bit, #hi-bit, #A mod 2).  Those parities for ALL windows of
//     one k are three XORs of a prefix-parity mask with itself shifted by k, so
//     the 8 bucket sizes are 8 popcounts and max-bucket >= MAX (kmer.cpp:2202).
//     A (segment,k) whose bound is below LOW_BASELINE*COUNT can never be
//     accepted by the selection loops (kmer.cpp:2221-2258); everything else is
//     a "candidate".  Reads with no candidate (~98.5 % of WGS-like input) are
//     finished here.  The bound is sound by construction: it never drops a k.
//
//  2. exact_kernel       one WAVEFRONT (64 lanes) per surviving read.  Restates
//     k_mer_check / k_mer_target / buffer_task* exactly, but only for candidate
//     k: the same bucket bound for all k at once (lane = k) prunes against the running
//     thresholds; survivors are split into runs of adjacent same-class windows (Lemma A),
//     one canonical rotation per run, runs merged per class (ballot + DPP sum); MAX_SEQ's
//     "first class to reach the maximum" tie-break (strict '<' at kmer.cpp:2202) is the
//     class whose last window comes first.  Histograms go through a wave-private LDS count
//     cache into a device-resident open-addressing table with 64-bit CAS keys.
//
// No CUDA shims, no dual paths: HIP for gfx950 only.
//
// One translation unit; the device code lives in kernels/*.inc, included below in dependency order:
//   helpers.inc        funnel shifts, DPP wave reductions, read descriptors, plane loads
//   prefilter.inc      filter_kernel<NW> (general path + uniform-geometry fast path)
//   count_table.inc    table_add / table_add_wide / spill log
//   exact_core.inc     LDS working set, eval_k / eval_runs (Lemma A), lane_bounds, decide, emit_k, run_short, run_segment
//   decide_group.inc   decide_group (four segments in lock step, 16 lanes each), run_short_group
//   driver_long.inc    run_long        driver_pair.inc   run_pair
//   exact_kernel.inc   exact_kernel<NW, MODE, WT>
//   table_kernels.inc  add-rows / compaction kernels      synth_kernels.inc  workload generators
// The launchers (host code) follow the includes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "trew_common.hpp"
#include "trew_launch.hpp"
#include "trew_synth.hpp"

#ifndef TREW_FILTER_THREADS
#define TREW_FILTER_THREADS 256
#endif

namespace trew {

typedef unsigned __int128 u128;  // 2k-bit words for k in (32, 64] (k_mer_check_128, kmer.cpp:2346-2547)

#include "kernels/helpers.inc"
#include "kernels/prefilter.inc"
#include "kernels/count_table.inc"
#include "kernels/exact_core.inc"
#include "kernels/decide_group.inc"
#include "kernels/driver_long.inc"
#include "kernels/driver_pair.inc"
#include "kernels/exact_kernel.inc"
#include "kernels/g1_compat.inc"
#include "kernels/table_kernels.inc"
#include "kernels/synth_kernels.inc"

// ------------------------------------------------------------------ launchers
int pick_nw(u32 max_seg_len) {
    if (max_seg_len <= 95) return 3;
    if (max_seg_len <= 159) return 5;
    if (max_seg_len <= 319) return 10;
    return 32;
}

// Pass thresholds of the uniform-geometry fast path for one batch geometry: table[slot * kThrRow + k - 1] = {ithr, jthr}
// (kThrTableBytes in all: the int2 rows, then the joint loop's int4 rows).
// Host side (single-precision multiply, the same IEEE operation the general path performs on the device).
void fill_thresholds(const DevParams &P, u32 uniform_length, int2 *table) {
    const int nslots = mode_slots(P.mode);
    u32 max_seg = 0;  // the kernel instantiation launch_filter picks for this geometry
    for (int slot = 0; slot < nslots; slot++) {
        const Segment sg = get_segment(P.mode, slot, uniform_length, uniform_length, P.min_mer, P.max_mer, P.slice_len);
        if (sg.valid) max_seg = std::max(max_seg, sg.len);
    }
    const int nw = pick_nw(max_seg);
    for (int slot = 0; slot < kMaxSlots; slot++) {
        const Segment sg = get_segment(P.mode, slot, uniform_length, uniform_length, P.min_mer, P.max_mer, P.slice_len);
        for (int k = 1; k <= kThrRow; k++) {
            const int W = (int) sg.len - k + 1;
            int2 th;
            th.x = kThrNever;  // nothing passes
            th.y = -1;
            if (slot < nslots && sg.valid && W > 0) {
                const volatile float prod = (float) W * P.lowf;  // volatile: no contraction, no extended precision
                // a class needs ithr of the W windows; of the Weff windows the kernel looks at it then has at least ithr - (W - Weff)
                const int Weff = uni_windows(nw, (int) sg.len, k);
                th.x = (int) floorf(prod) + 1 - (W - Weff);
                th.y = Weff - th.x;
            }
            table[slot * kThrRow + k - 1] = th;
        }
    }
    // Joint rows (row kMaxSlots + p for the halves 2p, 2p + 1 of a read of odd length, where the right half is one base longer
    // than the left): the prefilter judges both halves in ONE k loop with the left half's geometry -- the right half by its
    // first len_A bases, B'.  The windows of B' are the first W_A of B's W_B = W_A + 1 windows, so a class with ithr_B of B's
    // windows has at least ithr_B - (W_B - Weff) among the Weff windows the kernel looks at; one threshold serves both halves:
    // the smaller of the two (the right half's), which keeps the filter sound and costs the left half at most one count.
    for (int p = 0; p < 2; p++) {
        const Segment a = get_segment(P.mode, 2 * p, uniform_length, uniform_length, P.min_mer, P.max_mer, P.slice_len);
        const Segment b = get_segment(P.mode, 2 * p + 1, uniform_length, uniform_length, P.min_mer, P.max_mer, P.slice_len);
        const bool ok = 2 * p + 1 < nslots && a.valid && b.valid && b.len == a.len + 1;
        for (int k = 1; k <= kThrRow; k++) {
            int2 th;
            th.x = kThrNever;
            th.y = -1;
            const int WA = (int) a.len - k + 1, WB = WA + 1;
            if (ok && WA > 0) {
                const int Weff = uni_windows(nw, (int) a.len, k);
                const volatile float pa = (float) WA * P.lowf, pb = (float) WB * P.lowf;
                const int xa = (int) floorf(pa) + 1 - (WA - Weff), xb = (int) floorf(pb) + 1 - (WB - Weff);
                th.x = std::min(xa, xb);
                th.y = Weff - th.x;
            }
            table[(kMaxSlots + p) * kThrRow + k - 1] = th;
        }
    }
    // The joint loop's own rows (filter_halves_uni): per k {-2 ithr, -ithr, 3 ithr - jthr - 1, ithr} of the row it used to read --
    // the starting values of its popcount chains and the constant of its bucket-00 test (halves_signs).
    int4 *joint = (int4 *) (table + kThrRows * kThrRow);
    for (int r = 0; r < kThrJointRows; r++) {
        const int2 *src = table + (r < 2 ? 2 * r : kMaxSlots + (r - 2)) * kThrRow;
        for (int k = 1; k <= kThrRow; k++) {
            const int2 th = src[k - 1];
            joint[r * kThrRow + k - 1] = make_int4(-2 * th.x, -th.x, 3 * th.x - th.y - 1, th.x);
        }
    }
}

hipError_t launch_filter(hipStream_t st, u32 n_cu, u32 max_seg_len, const DevParams &P, const DevBatch &B, u32 *wl, u32 *wl_count,
                         u32 wl_cap, u64 *dbg_masks, int dbg_slots, u32 *diag, const int2 *d_thr) {
    const int nw = pick_nw(max_seg_len);
    const int max_seg = (int) std::min<u32>(max_seg_len, (u32) (32 * nw - 1));
    if (B.n_units == 0) return hipSuccess;
    typedef void (*kern_t)(DevParams, DevBatch, u32 *, u32 *, u32, u64 *, int, int, u32 *, const int2 *);
    const kern_t fn = nw == 3 ? filter_kernel<3> : nw == 5 ? filter_kernel<5> : nw == 10 ? filter_kernel<10> : filter_kernel<32>;
    const u32 threads = kFilterThreads;
    // Persistent blocks that pull chunks of reads from the device queue; as many as are resident at once (78 VGPRs -> 6 waves
    // per SIMD for 150-bp reads).  (Twice that number -- a relic of the static, grid-strided partition -- only put blocks in
    // the dispatcher's queue that find the chunk queue empty, and took wave slots from a co-resident exact kernel of the other
    // stream: step 1.038 -> 1.018 ms with the resident number, profiles/r03/README.md.)
    static thread_local kern_t cached_fn = nullptr;
    static thread_local int cached_per_cu = 0;
    int per_cu = cached_per_cu;
    if (cached_fn != fn) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) fn, (int) threads, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (const char *e = getenv("TREW_FILTER_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(e));  // experiments only
        cached_fn = fn;
        cached_per_cu = per_cu;
    }
    u64 blocks = (B.n_units + threads - 1) / threads;
    const u64 persistent = (u64) n_cu * (u64) per_cu;
    if (blocks > persistent) blocks = persistent;
    hipLaunchKernelGGL(fn, dim3((u32) blocks), dim3(threads), 0, st, P, B, wl, wl_count, wl_cap, dbg_masks, dbg_slots, max_seg, diag, d_thr);
    return hipGetLastError();
}

hipError_t fallback_counters_read(u32 *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fallback), sizeof(u32) * kFallbackWords); }
hipError_t fallback_counters_clear(hipStream_t st) {  // in stream order
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_fallback));
    return e != hipSuccess ? e : hipMemsetAsync(p, 0, sizeof(u32) * kFallbackWords, st);
}

u32 exact_lds_bytes_host(u32 cap, u32 rawwords, u32 wordbytes) { return exact_lds_bytes(cap, rawwords, wordbytes); }

// TREW_FLAG_COMPAT_G1: behind the exact kernel of a pair batch, on the same stream (see kernels/g1_compat.inc)
hipError_t launch_g1_apply(hipStream_t st, const DevTable &T, const DevG1 &G, const DevBatch &B, int min_mer, const trew_hip_row *carry_in,
                           trew_hip_row *carry_out, u32 carry_cap) {
    hipLaunchKernelGGL(g1_apply_kernel, dim3(256), dim3(256), 0, st, T, G, B, min_mer, carry_in, carry_out, carry_cap);
    hipLaunchKernelGGL(g1_finish_kernel, dim3(1), dim3(1), 0, st, G);
    return hipGetLastError();
}

hipError_t launch_exact(hipStream_t st, u32 n_cu, u64 n_units, const DevParams &P, const DevBatch &B, const DevTableG1 &T,
                        const u32 *wl, u32 *wl_count, u32 *wl_count_next, u32 wl_cap, const SegResults &R, u32 cap, u32 rawwords,
                        u32 max_seg_len, bool share) {
    // lane_bounds needs every staged segment (< cap) to fit its NW words
    const bool wide = P.max_mer > 32;  // 128-bit words, k_mer_check_128 (kmer.cpp:100, 180)
    const u32 lds = exact_lds_bytes(cap, rawwords, wide ? 16u : 8u);
    // max_seg_len = longest segment decide() is ever called on (halves, whole-read check, slices)
    // (k = 64 itself has no lane bound -- decide() walks its windows -- but every smaller k of a MAX_MER = 64 run does)
    // long mode: the kernel is built for the regular slice (SLICE_LENGTH bases); its driver decides the one longer middle slice
    // without lane bounds (run_long), so that slice does not set the register budget
    const u32 bound_len = P.mode == TREW_MODE_LONG ? (u32) P.slice_len : max_seg_len;
    const int nw = (P.flags & TREW_FLAG_NO_FILTER) ? 0 : (bound_len <= 95 ? 3 : bound_len <= 159 ? 5 : bound_len <= 319 ? 10 : 0);
    // one block = one wave; fill the chip exactly once (persistent, self-scheduling waves)
    typedef void (*kern_t)(DevParams, DevBatch, DevTableG1, const u32 *, u32 *, u32 *, u32, SegResults, u32, u32);
    kern_t fn = nullptr;
#define TREW_PICK_MODE(NWV, WTV)                                                      \
    switch (P.mode) {                                                                 \
    case TREW_MODE_SHORT: fn = exact_kernel<NWV, TREW_MODE_SHORT, WTV>; break;        \
    case TREW_MODE_PAIR: fn = exact_kernel<NWV, TREW_MODE_PAIR, WTV>; break;          \
    case TREW_MODE_LONG: fn = exact_kernel<NWV, TREW_MODE_LONG, WTV>; break;          \
    default: fn = exact_kernel<NWV, TREW_MODE_SEGMENT, WTV>; break;                   \
    }
    if (!wide) {
        switch (nw) {
        case 3: TREW_PICK_MODE(3, u64) break;
        case 5: TREW_PICK_MODE(5, u64) break;
        case 10: TREW_PICK_MODE(10, u64) break;
        default: TREW_PICK_MODE(0, u64) break;
        }
    } else {
        switch (nw) {
        case 3: TREW_PICK_MODE(3, u128) break;
        case 5: TREW_PICK_MODE(5, u128) break;
        case 10: TREW_PICK_MODE(10, u128) break;
        default: TREW_PICK_MODE(0, u128) break;
        }
    }
#undef TREW_PICK_MODE
    // the occupancy query is not free: remember the last answer
    // (per host thread: the packer threads of the CLI submit concurrently on their own slots)
    static thread_local kern_t cached_fn = nullptr;
    static thread_local u32 cached_lds = 0;
    static thread_local int cached_per_cu = 0;
    int per_cu = 8;
    if (cached_fn == fn && cached_lds == lds) {
        per_cu = cached_per_cu;
    } else {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) fn, 64, lds) != hipSuccess || per_cu < 1) per_cu = 8;
        cached_fn = fn;
        cached_lds = lds;
        cached_per_cu = per_cu;
    }
    per_cu = per_cu > 32 ? 32 : per_cu;
    // share: another slot's batch is in flight, i.e. a prefilter will run beside this kernel.  Both are persistent: whoever
    // starts first holds every wave slot until its queue runs dry, and the other only fills the tail.  Half the slots each
    // keeps both resident from start to end (measured, 10 M reads a step, two streams: 1.048 -> 0.989 ms; pair mode
    // 11.14 -> 11.04 ms; long reads, where the prefilter is a quarter of the work, are better off with the whole chip:
    // 3.60 against 3.69 ms -- profiles/r03/README.md).
    // Round 4: with the pair kernel a third shorter the split stopped paying there too (50 M pairs: 11.15 ms a step with half the
    // slots, 9.44 with all of them, 9.38 on one stream -- the kernels of the two slots run one after the other either way).
    if (share && P.mode == TREW_MODE_SHORT) per_cu = std::max(1, per_cu / 2);
    if (const char *e = getenv("TREW_EXACT_WAVES_PER_CU")) per_cu = std::max(1, atoi(e));  // experiments only
    // Self-scheduling waves: any grid size is correct, it only has to be large enough to keep the chip busy.  A big
    // batch gets every resident wave slot; a small one (the CLI's ~10^5-read batches, of which 1-2 % survive the
    // prefilter) one wave per 16 units, so that a launch does not start thousands of waves that find the queue empty.
    const u64 full = (u64) n_cu * (u64) per_cu;
    const u32 grid = (u32) std::min<u64>(full, std::max<u64>(n_units / 16, 64));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, st, P, B, T, wl, wl_count, wl_count_next, wl_cap, R, cap, rawwords);
    return hipGetLastError();
}

// flags: device scratch (kRowFlagWords words, zeroed by the caller on the same stream) or nullptr when the rows were
// validated on the host.  With flags the check kernel runs first and the add kernel behind it does nothing if it found a bad row.
hipError_t launch_add_rows(hipStream_t st, const DevTable &T, const trew_hip_row *d_rows, u64 n, u32 *d_flags) {
    if (n == 0) return hipSuccess;
    const dim3 grid((u32) ((n + 255) / 256));
    if (d_flags) hipLaunchKernelGGL(table_check_rows_kernel, grid, dim3(256), 0, st, d_rows, n, d_flags);
    hipLaunchKernelGGL(table_add_rows_kernel, grid, dim3(256), 0, st, T, d_rows, n, (const u32 *) d_flags);
    return hipGetLastError();
}

// the gather buffer of the cross-GPU exchange: n_slices x (1 + slice_rows) rows, headers in row 0 of each slice
hipError_t launch_add_gathered(hipStream_t st, const DevTable &T, const trew_hip_row *d_buf, u32 n_slices, u32 own, u64 slice_rows, u32 *d_flags) {
    const u64 total = (u64) n_slices * (slice_rows + 1ull);
    if (total == 0) return hipSuccess;
    const dim3 grid((u32) ((total + 255) / 256));
    hipLaunchKernelGGL(table_check_gathered_kernel, grid, dim3(256), 0, st, d_buf, n_slices, own, slice_rows, d_flags);
    hipLaunchKernelGGL(table_add_gathered_kernel, grid, dim3(256), 0, st, T, d_buf, n_slices, own, slice_rows, (const u32 *) d_flags);
    return hipGetLastError();
}

hipError_t launch_pack_ascii(hipStream_t st, const unsigned char *d_bases, const u32 *d_byte_offsets, const u32 *d_lengths, const u32 *d_word_offsets,
                             u32 uniform_length, u64 n_reads, u64 n_triples, u32 *d_words) {
    if (n_triples == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_ascii_kernel, dim3((u32) ((n_triples + 255) / 256)), dim3(256), 0, st, d_bases, d_byte_offsets, d_lengths, d_word_offsets, uniform_length,
                       n_reads, n_triples, d_words);
    return hipGetLastError();
}

hipError_t launch_compact(hipStream_t st, const DevTable &T, u64 n_slots, u32 wide_log2_slots, int table, trew_hip_row *d_rows, u64 cap,
                          unsigned long long *d_n) {
    const u64 total = n_slots + (1ull << wide_log2_slots);
    hipLaunchKernelGGL(table_compact_kernel, dim3((u32) ((total + 255) / 256)), dim3(256), 0, st, T, n_slots, table, d_rows, cap, d_n);
    return hipGetLastError();
}

// header + spill log of a rank's exchange slice, behind launch_compact on the same stream (see table_slice_finish_kernel)
hipError_t launch_slice_finish(hipStream_t st, trew_hip_row *d_slice, u64 slice_rows, const unsigned long long *d_n, const trew_hip_row *d_spill_rows,
                               const u32 *d_spill_n, u32 spill_cap) {
    hipLaunchKernelGGL(table_slice_finish_kernel, dim3((spill_cap + 255u) / 256u), dim3(256), 0, st, d_slice, slice_rows, d_n, d_spill_rows, d_spill_n, spill_cap);
    return hipGetLastError();
}

hipError_t launch_synth_short(hipStream_t st, u64 seed, u64 first, u64 n, u32 len, u32 *d_words) {
    const u64 total = n * ((len + 31u) >> 5);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_short_kernel, dim3((u32) ((total + 255) / 256)), dim3(256), 0, st, seed, first, n, len, d_words);
    return hipGetLastError();
}

hipError_t launch_synth_long(hipStream_t st, u64 seed, u64 first, u64 n, const u32 *d_qtable, const u32 *d_offsets, u32 *d_words) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_long_kernel, dim3((u32) n), dim3(64), 0, st, seed, first, n, d_qtable, d_offsets, d_words);
    return hipGetLastError();
}

hipError_t launch_synth_pair(hipStream_t st, u64 seed, u64 first, u64 n, u32 len, u32 *d_words) {
    const u64 total = n * 2ull * ((len + 31u) >> 5);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_pair_kernel, dim3((u32) ((total + 255) / 256)), dim3(256), 0, st, seed, first, n, len, d_words);
    return hipGetLastError();
}

}  // namespace trew
