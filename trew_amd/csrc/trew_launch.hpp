// trew_launch.hpp -- host-callable launchers implemented in trew_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "trew_common.hpp"

namespace trew {

int pick_nw(u32 max_seg_len);

hipError_t launch_filter(hipStream_t st, u32 n_cu, u32 max_seg_len, const DevParams &P, const DevBatch &B, u32 *wl, u32 *wl_count,
                         u32 wl_cap, u64 *dbg_masks, int dbg_slots, u32 *diag, const int2 *d_thr);
void fill_thresholds(const DevParams &P, u32 uniform_length, int2 *table);  // kThrTableBytes
hipError_t launch_exact(hipStream_t st, u32 n_cu, u64 n_units, const DevParams &P, const DevBatch &B, const DevTableG1 &T,
                        const u32 *wl, u32 *wl_count, u32 *wl_count_next, u32 wl_cap, const SegResults &R, u32 cap, u32 rawwords,
                        u32 max_seg_len, bool share);  // share: leave half of the wave slots to a prefilter on another stream
u32 exact_lds_bytes_host(u32 cap, u32 rawwords, u32 wordbytes);
hipError_t launch_g1_apply(hipStream_t st, const DevTable &T, const DevG1 &G, const DevBatch &B, int min_mer, const trew_hip_row *carry_in,
                           trew_hip_row *carry_out, u32 carry_cap);
hipError_t fallback_counters_read(u32 *out);  // kFallbackWords words of the current device
hipError_t fallback_counters_clear(hipStream_t st);  // queued on st
hipError_t launch_add_rows(hipStream_t st, const DevTable &T, const trew_hip_row *d_rows, u64 n, u32 *d_flags);
hipError_t launch_add_gathered(hipStream_t st, const DevTable &T, const trew_hip_row *d_buf, u32 n_slices, u32 own, u64 slice_rows, u32 *d_flags);
hipError_t launch_pack_ascii(hipStream_t st, const unsigned char *d_bases, const u32 *d_byte_offsets, const u32 *d_lengths, const u32 *d_word_offsets,
                             u32 uniform_length, u64 n_reads, u64 n_triples, u32 *d_words);
hipError_t launch_compact(hipStream_t st, const DevTable &T, u64 n_slots, u32 wide_log2_slots, int table, trew_hip_row *d_rows, u64 cap,
                          unsigned long long *d_n);
hipError_t launch_slice_finish(hipStream_t st, trew_hip_row *d_slice, u64 slice_rows, const unsigned long long *d_n, const trew_hip_row *d_spill_rows,
                               const u32 *d_spill_n, u32 spill_cap);
hipError_t launch_synth_short(hipStream_t st, u64 seed, u64 first, u64 n, u32 len, u32 *d_words);
hipError_t launch_synth_long(hipStream_t st, u64 seed, u64 first, u64 n, const u32 *d_qtable, const u32 *d_offsets, u32 *d_words);
hipError_t launch_synth_pair(hipStream_t st, u64 seed, u64 first, u64 n, u32 len, u32 *d_words);

}  // namespace trew
