// trew_common.hpp -- types shared by the HIP kernels and the C-ABI host layer.
//
// MI355X-native restatement of the hot path of Chemical118/TREW src/kmer.cpp.
// Nothing here is derived from the reference's data structures: per-thread
// direct-address counters / hash maps (ThreadData, kmer.h:132-154) are replaced
// by registers and LDS, the six ResultMaps (kmer.h:79-81) by one device-resident
// open-addressing table with 64-bit CAS keys.
#pragma once
#include <stdint.h>

#include "../../include/trew_hip.h"

namespace trew {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kMaxSlots = 6;        // segments per unit: short 3, pair 6, long 2, segment 1
constexpr int kMaxSegBases = 1023;  // longest segment any kernel accepts (short mode rejects reads > 1000, kmer.cpp:1006-1009)
constexpr int kTablePartBits = 9;   // low word bits that select the table partition (see table_add)
constexpr int kThrRows = kMaxSlots + 2;  // rows of the threshold table: one per slot + one per pair of halves of unequal length (joint rows, fill_thresholds)
constexpr int kThrJointRows = 4;    // behind those, four rows of int4 for the loop that judges both halves of a read at once: halves 0/1 and 2/3 of equal length, then of unequal length
constexpr int kThrRow = 66;         // threshold-table entries per slot: k = 1..64, one of read-ahead padding, one to keep rows 16-byte aligned
constexpr size_t kThrTableBytes = (size_t) kThrRows * kThrRow * 8 + (size_t) kThrJointRows * kThrRow * 16;
constexpr int kThrNever = 1 << 20;  // a threshold nothing reaches (small enough to be doubled and tripled in the joint rows)

struct DevParams {
    int min_mer, max_mer;
    double low, high;
    float lowf;  // conservative float lower bound of `low` used by the prefilter
    int slice_len;
    int mode;
    u32 flags;
};

struct DevBatch {
    const u32 *words;
    const u32 *offsets;  // nullable
    const u32 *lengths;  // nullable
    u32 uniform_length;
    u32 uniform_stride;
    u64 n_reads;
    u64 n_units;  // reads, or pairs in pair mode
};

// the worklist is a plain array of unit indices (reads / pairs) that survived the prefilter
// One counter block per launch (a slot owns two and alternates, trew_capi.cpp): word 0 = worklist size, then eight 32-word
// lines: word 0 of line s = the exact kernel's queue head of shard s, word kChunkCounterWord = the prefilter's chunk counter of shard s.
constexpr int kChunkCounterWord = 16;

// words of DevTable::overflow (one 64-byte line of device counters, cleared by trew_hip_reset_tables)
enum {
    kDiagOverflow = 0,     // a row found the table AND the spill log full: counts were lost, collect fails
    kDiagWorklistDrop = 1, // the prefilter had more survivors than the worklist holds (cannot happen: cap = batch capacity)
    kDiagIntentDrop = 2,   // the pair driver logged more than 32 deferred emissions for one pair (cannot happen: <= 20)
    kDiagInserted = 3,     // keys inserted into the narrow table (occupancy, trew_hip_table_pressure)
    kDiagInsertedWide = 4, // keys inserted into the wide table
    kDiagKernarg = 5,      // TREW_FLAG_DEBUG_POISON_LDS runs: the table descriptor read through the kernarg pointer differed from the by-value argument
    kDiagSpillRows = 6,    // rows in the spill log (DevWide::spill_n points here: one copy reads the whole fill state)
    kDiagG1Drop = 7,       // TREW_FLAG_COMPAT_G1: the stale-row log or its carry was full (collect fails)
    kDiagWords = 16
};

// How often the kernels took their rare fall-back paths (trew_hip_debug_counters).  A module-level device array rather than
// words of DevTable::overflow: decide() and eval_runs() have no table pointer, and the exact kernels have no SGPRs to spare
// for one -- the address of a __device__ variable is a literal.  One array per device, shared by the contexts on it.
enum {
    kFallbackStrictRerun = 0,   // decide(): a speculative skip failed its check and the segment was decided again, every k counted
    kFallbackWindows = 1,       // eval_runs(): more than 64 runs, classes counted per window (eval_k_windows)
    kFallbackWideSpin = 2,      // table_add_wide(): gave up waiting for a slot's ready bit (a duplicate slot may follow; collect merges)
    kFallbackGroupPunt = 3,     // decide_group(): a row gave its segment back (an N, too many heavy k, a failed skip check) and decide() took it
    kFallbackGroupRouted = 4,   // run_short_group(): a read routed and recorded by run_short_routed instead of in row space
    kFallbackGroupTarget = 5,   // run_short_group(): k_mer_target of a read counted by target() (more than 16 runs in the whole read)
    kFallbackWords = 8
};

// device scratch of the row-adding entry points (table_check_rows_kernel & co): validation is a pass of its own, so an add is all or nothing
enum { kRowFlagBad = 0, kRowFlagOverflow = 1, kRowFlagMaxLo = 2, kRowFlagMaxHi = 3, kRowFlagWords = 4 };

struct DevTable {
    u64 *keys;    // 0 = empty
    u64 *counts;
    u32 log2_part_slots;  // slots per partition = 1 << log2_part_slots; 512 partitions
    u32 *overflow;        // kDiagWords counters, see above
    const struct DevWide *wide;  // device-resident descriptor of the wide-entry table (k > 32)
};

// TREW_FLAG_COMPAT_G1 (pair mode): what the whole-read block of a pair recorded into temp_result_left stays there in the
// reference's 64-bit branch and is added again by the next pair.  The exact kernel logs those rows and every pair's
// chain flags; g1_apply_kernel adds them again after the batch's exact kernel, rows of the batch's last pair travel to the next
// batch in a carry buffer.  Lives behind the DevTable in the exact kernels' descriptor argument (DevTableG1, defined below DevWide).
struct DevG1 {
    trew_hip_row *log;          // {k, table = forward_high | forward_low, word_lo, word_hi = pair index in the batch, count}
    u32 *counters;              // [0] rows in log, [1] rows in the carry read by this batch, [2] rows in the carry it writes
    unsigned char *pair_flags;  // per pair: bit b = all four segments chained for baseline b (kmer.cpp:378, 389);
                                // bit 2 + b = the whole-read block's `both` condition (kmer.cpp:487, 494); zeroed per batch
    u32 log_cap;
};


// wide entries (k in (32, 64], 128-bit words): tag / word halves / count per slot.  Kept behind a
// pointer so that DevTable stays small enough to travel in registers.
// spill_*: append-only log of (table, k, word, count) rows that found their table partition (or the
// wide table) full; trew_hip_collect merges it, so a skewed key distribution degrades gracefully
// instead of failing.  overflow (DevTable) is only raised when the log itself is full.
struct DevWide {
    u64 *wtag, *wlo, *whi, *wcount;
    u32 wide_log2_slots;
    u32 spill_cap;
    trew_hip_row *spill_rows;
    u32 *spill_n;
    u32 spin_limit;  // polls of a claimed slot's ready bit before table_add_wide gives up on it (0 with TREW_FLAG_DEBUG_WIDE_NO_WAIT)
};

struct DevTableG1 {
    DevTable t;
    DevG1 g;
    DevWide w;  // the wide-table descriptor by value as well: the exact kernels read it where t.wide would point (load_table)
};

// per-read outputs of TREW_MODE_SEGMENT
struct SegResults {
    int32_t *k_high;
    int32_t *k_low;
    u64 *seq_high;     // low 64 bits of MAX_SEQ at k_high
    u64 *seq_low;
    u64 *seq_high_hi;  // high 64 bits (k > 32)
    u64 *seq_low_hi;
};

struct Segment {
    u32 mate;   // 0 = first read of the unit, 1 = second (pair mode)
    u32 start;  // first base
    u32 len;    // bases
    int kmin, kmax;
    bool valid;
};

#if defined(__HIPCC__)
#define TREW_HD __host__ __device__
#else
#define TREW_HD
#endif

// Segment geometry of every per-read driver of the reference, as a pure
// function of the read length(s).  slot numbering:
//   short  (buffer_task, kmer.cpp:115-171):       0 left half, 1 right half, 2 whole read
//   pair   (buffer_task_pair, kmer.cpp:333-340, 467-480): 0 R1-left 1 R1-right 2 R2-right 3 R2-left 4 R1 whole 5 R2 whole
//   long   (buffer_task_long, kmer.cpp:790-798, 836-838): 0 first slice, 1 last slice (interior slices on demand)
//   segment (k_mer_check, kmer.h:232):            0 whole read
TREW_HD inline Segment get_segment(int mode, int slot, u32 n1, u32 n2, int MIN_MER, int MAX_MER, int SLICE) {
    Segment s;
    s.mate = 0;
    s.start = 0;
    s.len = 0;
    s.kmin = 1;
    s.kmax = 0;
    s.valid = false;
    auto imin = [](int a, int b) { return a < b ? a : b; };
    auto imax = [](int a, int b) { return a > b ? a : b; };
    if (mode == TREW_MODE_SHORT) {
        int n = (int) n1;
        if (2 * MIN_MER > n) return s;
        if (slot <= 1) {
            if (4 * MIN_MER > n) return s;
            s.kmin = MIN_MER;
            s.kmax = imin(n / 4, MAX_MER);
            if (slot == 0) {
                s.start = 0;
                s.len = (u32) (n / 2);
            } else {
                s.start = (u32) (n - (n + 1) / 2);
                s.len = (u32) ((n + 1) / 2);
            }
            s.valid = true;
        } else if (slot == 2) {
            if (4 * MAX_MER <= n) return s;
            s.kmin = imax(n / 4 + 1, MIN_MER);
            s.kmax = imin(n / 2, MAX_MER);
            s.start = 0;
            s.len = (u32) n;
            s.valid = s.kmin <= s.kmax;
        }
    } else if (mode == TREW_MODE_PAIR) {
        int a = (int) n1, b = (int) n2;
        int n = imin(a, b);
        if (2 * MIN_MER > n) return s;
        if (slot <= 3) {
            if (4 * MIN_MER > n) return s;
            s.kmin = MIN_MER;
            s.kmax = imin(n / 4, MAX_MER);
            if (slot == 0) {
                s.mate = 0; s.start = 0; s.len = (u32) (a / 2);
            } else if (slot == 1) {
                s.mate = 0; s.start = (u32) (a - (a + 1) / 2); s.len = (u32) ((a + 1) / 2);
            } else if (slot == 2) {
                s.mate = 1; s.start = (u32) (b - (b + 1) / 2); s.len = (u32) ((b + 1) / 2);
            } else {
                s.mate = 1; s.start = 0; s.len = (u32) (b / 2);
            }
            s.valid = true;
        } else if (slot <= 5) {
            if (4 * MAX_MER <= n) return s;
            s.kmin = imax(n / 4 + 1, MIN_MER);
            s.kmax = imin(n / 2, MAX_MER);
            s.mate = (u32) (slot - 4);
            s.start = 0;
            s.len = slot == 4 ? (u32) a : (u32) b;
            s.valid = s.kmin <= s.kmax;
        }
    } else if (mode == TREW_MODE_LONG) {
        int len = (int) n1;
        int snum = len / SLICE;
        if (snum == 0 || slot > 1) return s;
        int mid = (snum + 1) / 2;
        int bonus = len % SLICE;
        s.kmin = MIN_MER;
        s.kmax = MAX_MER;
        if (slot == 0) {
            s.start = 0;
            s.len = (u32) (SLICE + (1 == mid ? bonus : 0));
        } else {
            int sl = SLICE + (snum == mid ? bonus : 0);
            s.start = (u32) (len - sl);
            s.len = (u32) sl;
        }
        s.valid = true;
    } else {  // TREW_MODE_SEGMENT
        if (slot != 0 || n1 == 0) return s;
        s.kmin = MIN_MER;
        s.kmax = MAX_MER;
        s.start = 0;
        s.len = n1;
        s.valid = true;
    }
    return s;
}

// Windows the prefilter's uniform-geometry fast path LOOKS AT for (segment length L, k) in an nw-word kernel: all COUNT = L-k+1
// of them, except that the 3-word kernel stops at the first 64 when there are 65..kUniSubsetMax (a sound subset bound, see
// filter_k_uni).  fill_thresholds (host) and the kernel both go by this function.
#ifndef TREW_AB_NO_SUBSET
constexpr int kUniSubsetMax = 72;
#else
constexpr int kUniSubsetMax = 64;  // A/B builds: no subset range
#endif
TREW_HD inline int uni_windows(int nw, int L, int k) {
    const int W = L - k + 1;
#ifdef TREW_AB_NO_CONTAINER
    (void) nw;
    return W;
#else
    return (nw == 3 && W > 64 && W <= kUniSubsetMax) ? 64 : W;
#endif
}

TREW_HD inline int mode_slots(int mode) {
    return mode == TREW_MODE_SHORT ? 3 : mode == TREW_MODE_PAIR ? 6 : mode == TREW_MODE_LONG ? 2 : 1;
}

}  // namespace trew
