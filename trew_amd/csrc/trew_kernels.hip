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

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ u32 alignbit(u32 hi, u32 lo, u32 sh) {
    return __builtin_amdgcn_alignbit(hi, lo, sh);  // ((hi:lo) >> (sh & 31)) & 0xffffffff
}
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
// popc(x) + acc in one instruction
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) {
    u32 d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}
__device__ __forceinline__ u32 rfl(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 rfl64(u64 v) {
    u32 lo = __builtin_amdgcn_readfirstlane((u32) v);
    u32 hi = __builtin_amdgcn_readfirstlane((u32) (v >> 32));
    return ((u64) hi << 32) | lo;
}

// Arguments of a noinline device function arrive in VGPRs, so the compiler has to treat them as
// divergent: loops run on exec masks and address arithmetic on the VALU.  The values below are
// wave-uniform by construction; re-reading them through readfirstlane moves them (and everything
// derived from them) to the scalar unit.
__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename Tp>
__device__ __forceinline__ Tp *rfl_ptr(Tp *p) {
    return (Tp *) rfl64((u64) p);
}
__device__ __forceinline__ double rfl_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// wave max of a u32 (0 for idle lanes), every lane gets the result.  DPP row shifts and row
// broadcasts: 6 VALU instructions instead of 6 LDS-crossbar round trips (ds_bpermute).
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
    // max is idempotent, so overlapping shifts are fine: after row_shr 1,2,4,8 lane 15 of each row holds the row's max
    v = max(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, true));  // row_shr:1
    v = max(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, true));  // row_shr:2
    v = max(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, true));  // row_shr:4
    v = max(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, true));  // row_shr:8
    v = max(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1 and 3
    v = max(v, (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2 and 3
    return (u32) __builtin_amdgcn_readlane((int) v, 63);
}

// wave sum of a u32, every lane gets the result: Hillis-Steele inside each row of 16 lanes (row_shr
// 1, 2, 4, 8), then the row totals are passed on with the two row broadcasts; lane 63 holds the sum
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
    v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, true);
    v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, true);
    v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, true);
    v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, true);
    v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return (u32) __builtin_amdgcn_readlane((int) v, 63);
}

struct ReadRef {
    const u32 *w;  // first triple
    u32 len;       // bases
    u32 nw;        // triples
};
// one wave works on one read: its descriptor is wave-uniform
__device__ __forceinline__ ReadRef uni(ReadRef r) {
    r.w = rfl_ptr(r.w);
    r.len = rfl(r.len);
    r.nw = rfl(r.nw);
    return r;
}

__device__ __forceinline__ ReadRef get_read(const DevBatch &b, u64 r) {
    ReadRef x;
    u64 off = b.offsets ? (u64) b.offsets[r] : r * (u64) b.uniform_stride;
    x.len = b.lengths ? b.lengths[r] : b.uniform_length;
    x.w = b.words + off;
    x.nw = (x.len + 31u) >> 5;
    return x;
}

// 32*NW plane bits starting at base s of a read: lo/hi/nmask, bit i = base s+i.
template <int NW>
__device__ __forceinline__ void load_planes(const ReadRef &rd, u32 s, u32 (&lo)[NW], u32 (&hi)[NW], u32 (&nm)[NW]) {
    const u32 ws = s >> 5, bs = s & 31u;
    u32 c0 = 0, c1 = 0, c2 = 0;
    if (ws < rd.nw) {
        c0 = rd.w[3 * ws + 0];
        c1 = rd.w[3 * ws + 1];
        c2 = rd.w[3 * ws + 2];
    }
#pragma unroll
    for (int j = 0; j < NW; j++) {
        u32 n0 = 0, n1 = 0, n2 = 0;
        const u32 idx = ws + (u32) j + 1u;
        if (idx < rd.nw) {
            n0 = rd.w[3 * idx + 0];
            n1 = rd.w[3 * idx + 1];
            n2 = rd.w[3 * idx + 2];
        }
        lo[j] = alignbit(n0, c0, bs);
        hi[j] = alignbit(n1, c1, bs);
        nm[j] = alignbit(n2, c2, bs);
        c0 = n0;
        c1 = n1;
        c2 = n2;
    }
}

// ------------------------------------------------------------------ prefilter

// exclusive prefix parity of f over bits 0..32*NW-1: P[i] = XOR_{t<i} f[t]
template <int NW>
__device__ __forceinline__ void prefix_parity(const u32 (&f)[NW], u32 (&P)[NW]) {
    u32 carry = 0, prev_top = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        u32 x = f[j];
        x ^= x << 1;
        x ^= x << 2;
        x ^= x << 4;
        x ^= x << 8;
        x ^= x << 16;
        x ^= carry;  // carry = all-ones when the parity of all lower words is odd
        P[j] = (x << 1) | prev_top;
        prev_top = x >> 31;
        carry = 0u - prev_top;
    }
}

__device__ __forceinline__ u64 all_k_mask(int kmin, int kmax) {
    if (kmax < kmin) return 0;
    u64 hi = kmax >= 64 ? ~0ull : ((1ull << kmax) - 1ull);
    u64 lo = (1ull << (kmin - 1)) - 1ull;
    return hi & ~lo;
}

// word j of (x >> k), k = 32*WS + bs, zero-extended above word NW-1
template <int NW, int WS>
__device__ __forceinline__ u32 shr_word(const u32 (&x)[NW], int j, u32 bs) {
    const u32 lo = (j + WS < NW) ? x[j + WS < NW ? j + WS : 0] : 0u;
    const u32 hi = (j + WS + 1 < NW) ? x[j + WS + 1 < NW ? j + WS + 1 : 0] : 0u;
    return alignbit(hi, lo, bs);
}

// One k of the prefilter.  V holds V_k (windows with no N) on entry and V_{k+1} on exit.
// NWW = number of mask words that can still hold a window at this k (compile-time, so the
// per-word work has no scalar branches).
// Stage 1 uses two parities (4 buckets); the third parity (8 buckets) is evaluated only
// when some lane of the wave would otherwise get its FIRST candidate from the 4-bucket
// bound.  Lanes that already own a candidate keep the looser -- still sound -- verdict:
// the exact kernel prunes their extra candidates itself (lane_bounds).  The result of
// a lane never depends on its neighbours: 8-bucket max <= 4-bucket max.
// Measured on MI355X (tools/valu_rate.hip): v_alignbit / v_bcnt occupy a SIMD for ~4.2 cycles
// per wave-instruction, simple logic / add / v_bitop3 for ~2.3-2.7; the slow ones are what is
// minimised here (e.g. bucket 00 is COUNT minus the other three instead of a fourth popcount).
template <int NW, int WS, int NWW>
__device__ __forceinline__ void filter_k(const u32 (&P1)[NW], const u32 (&P2)[NW], const u32 (&P3)[NW],
                                         const u32 (&v1)[NW], u32 (&V)[NW], int k, float lowf, u32 &cand_lo, u32 &cand_hi) {
    const u32 bs = (u32) k & 31u;
    u32 c01 = 0, c10 = 0, c11 = 0, count = 0;
#pragma unroll
    for (int j = 0; j < NWW; j++) {
        const u32 F1 = P1[j] ^ shr_word<NW, WS>(P1, j, bs), F2 = P2[j] ^ shr_word<NW, WS>(P2, j, bs);
        const u32 v = V[j];
        const u32 a1 = v & F1;
        const u32 a11 = a1 & F2;
        count += __popc(v);
        c11 += __popc(a11);
        c10 += __popc(a1 ^ a11);
        c01 += __popc((v ^ a1) & F2);
    }
    const u32 c00 = count - c01 - c10 - c11;
    const u32 m4 = max(max(c00, c01), max(c10, c11));
    // MAX <= m4; MAX/COUNT >= LOW needs m4 >= LOW*COUNT > lowf*COUNT (lowf < LOW*(1-1e-6), COUNT > 0);
    // the strict '>' also rejects COUNT == 0 (m4 == 0)
    const float thr = (float) count * lowf;
    const bool pass4 = (float) m4 > thr;
    bool pass = pass4;
    const bool first = (cand_lo | cand_hi) == 0;
    if (__any(pass4 && first)) {
        u32 c001 = 0, c010 = 0, c011 = 0, c100 = 0, c101 = 0, c110 = 0, c111 = 0;
#pragma unroll
        for (int j = 0; j < NWW; j++) {
            const u32 F1 = P1[j] ^ shr_word<NW, WS>(P1, j, bs), F2 = P2[j] ^ shr_word<NW, WS>(P2, j, bs);
            const u32 F3 = P3[j] ^ shr_word<NW, WS>(P3, j, bs);
            const u32 v = V[j];
            const u32 a1 = v & F1, a0 = v ^ a1;
            const u32 a11 = a1 & F2, a10 = a1 ^ a11, a01 = a0 & F2, a00 = a0 ^ a01;
            const u32 b111 = a11 & F3, b101 = a10 & F3, b011 = a01 & F3, b001 = a00 & F3;
            c111 += __popc(b111);
            c110 += __popc(a11 ^ b111);
            c101 += __popc(b101);
            c100 += __popc(a10 ^ b101);
            c011 += __popc(b011);
            c010 += __popc(a01 ^ b011);
            c001 += __popc(b001);
        }
        const u32 c000 = count - c001 - c010 - c011 - c100 - c101 - c110 - c111;
        const u32 m8 = max(max(max(c000, c001), max(c010, c011)), max(max(c100, c101), max(c110, c111)));
        pass = first ? ((float) m8 > thr) : pass4;
    }
    if (k <= 32)
        cand_lo |= pass ? (1u << ((k - 1) & 31)) : 0u;
    else
        cand_hi |= pass ? (1u << ((k - 33) & 31)) : 0u;
    // V_{k+1} = V_k & (v1 >> k)
#pragma unroll
    for (int j = 0; j < NWW; j++) V[j] &= shr_word<NW, WS>(v1, j, bs);
}

// k range [klo, khi] split by the number of window words: windows i <= max_seg - k need
// ceil((max_seg - k + 1) / 32) words.  Recursion over NWW keeps every trip count static.
template <int NW, int WS, int NWW>
struct FilterRange {
    static __device__ __forceinline__ void run(const u32 (&P1)[NW], const u32 (&P2)[NW], const u32 (&P3)[NW], const u32 (&v1)[NW],
                                               u32 (&V)[NW], int klo, int khi, int max_seg, float lowf, u32 &clo, u32 &chi) {
        // k values whose windows need exactly NWW words (or more than NW: clamp) : max_seg+1-32*NWW < k <= max_seg+1-32*(NWW-1)
        int a = max_seg + 2 - 32 * NWW, b = max_seg + 1 - 32 * (NWW - 1);
        if (NWW == NW) a = klo;  // segments are never longer than the instantiation allows
        a = a < klo ? klo : a;
        b = b > khi ? khi : b;
        for (int k = a; k <= b; k++) filter_k<NW, WS, NWW>(P1, P2, P3, v1, V, k, lowf, clo, chi);
        FilterRange<NW, WS, NWW - 1>::run(P1, P2, P3, v1, V, klo, khi, max_seg, lowf, clo, chi);
    }
};
template <int NW, int WS>
struct FilterRange<NW, WS, 0> {
    static __device__ __forceinline__ void run(const u32 (&)[NW], const u32 (&)[NW], const u32 (&)[NW], const u32 (&)[NW], u32 (&)[NW],
                                               int, int, int, float, u32 &, u32 &) {}
};

// Candidate-k mask of one segment (bit k-1).  lo/hi/nm hold the segment's
// planes from bit 0; L <= 32*NW-1 bases.  gmin..gmax is the wave-uniform k
// loop (MIN_MER..MAX_MER); only k in [kmin,kmax] can become candidates.
// max_seg (wave-uniform) bounds L over the whole batch.
//
// Soundness: windows of one rotation class (kmer.cpp:1815-1823) have the same
// base composition, hence the same (#lo-bit, #hi-bit, #A) parities.  With
// P_b the exclusive prefix parity of feature b, the parity of window i is
// P_b[i] ^ P_b[i+k]; so the size of every parity bucket is one popcount and the
// largest bucket is an upper bound of K_MER_DATA_MAX (kmer.cpp:2202).
template <int NW>
__device__ __forceinline__ u64 filter_segment(const u32 (&lo)[NW], const u32 (&hi)[NW], const u32 (&nm)[NW], int L,
                                              int kmin, int kmax, int gmin, int gmax, int max_seg, float lowf) {
    u32 v1[NW], P1[NW], P2[NW], P3[NW], V[NW];
    {
        u32 f1[NW], f2[NW], f3[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
            int bits = L - 32 * j;
            u32 lm = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
            v1[j] = ~nm[j] & lm;  // base is A/C/G/T and inside the segment
            f1[j] = lo[j] & v1[j];
            f2[j] = hi[j] & v1[j];
            f3[j] = f1[j] & f2[j];
            V[j] = v1[j];
        }
        prefix_parity<NW>(f1, P1);
        prefix_parity<NW>(f2, P2);
        prefix_parity<NW>(f3, P3);
    }
    // V = V_gmin: windows of length gmin with no N (kmer.cpp:2190)
    for (int t = 1; t < gmin && t < 32; t++) {
#pragma unroll
        for (int j = 0; j < NW; j++) V[j] &= shr_word<NW, 0>(v1, j, (u32) t);
    }
    for (int t = 32; t < gmin && t < 64; t++) {
#pragma unroll
        for (int j = 0; j < NW; j++) V[j] &= shr_word<NW, 1>(v1, j, (u32) t & 31u);
    }
    u32 clo = 0, chi = 0;
    const int g31 = gmax < 31 ? gmax : 31;
    if (NW <= 5) {
        FilterRange<NW, 0, NW>::run(P1, P2, P3, v1, V, gmin, g31, max_seg, lowf, clo, chi);
        FilterRange<NW, 1, NW>::run(P1, P2, P3, v1, V, gmin > 32 ? gmin : 32, gmax < 63 ? gmax : 63, max_seg, lowf, clo, chi);
        FilterRange<NW, 2, NW>::run(P1, P2, P3, v1, V, gmin > 64 ? gmin : 64, gmax, max_seg, lowf, clo, chi);
    } else {
        // long segments: all words every k (static trip counts would multiply the code size)
        for (int k = gmin; k <= g31; k++) filter_k<NW, 0, NW>(P1, P2, P3, v1, V, k, lowf, clo, chi);
        for (int k = gmin > 32 ? gmin : 32; k <= (gmax < 63 ? gmax : 63); k++) filter_k<NW, 1, NW>(P1, P2, P3, v1, V, k, lowf, clo, chi);
        for (int k = gmin > 64 ? gmin : 64; k <= gmax; k++) filter_k<NW, 2, NW>(P1, P2, P3, v1, V, k, lowf, clo, chi);
    }
    // only k inside the segment's own range can be candidates
    return ((((u64) chi) << 32) | clo) & all_k_mask(kmin, kmax);
}


// ------------------------------------------------------------------ prefilter, uniform-geometry fast path
// Batches of equal-length reads (trew_hip_batch.uniform_length, the usual Illumina case) have the same
// segment geometry for every unit, so everything that depends only on (segment length, k) is wave-uniform:
// COUNT = L-k+1, the window mask of the last word and the pass threshold live in SGPRs / an LDS table
// instead of being recomputed per lane.  Reads with an N inside a segment do not fit that model (their
// COUNT differs); they are set aside in LDS and go through the general filter_segment, 256 at a time.
// Per k and word this leaves 2 alignbit + 2 xor + 1 and + 3 bcnt -- the kernel is VALU-issue bound
// (every wave64 VALU instruction occupies its SIMD16 for 4 cycles), so instruction count is time.
// The verdicts of the 64 lanes are kept as wave masks in SGPRs (hasm: lanes that already own a
// candidate, the return value: lanes passing at this k), so the bookkeeping per k is scalar work.
// thr_row[k-1] = {ithr, jthr} (kThrRow entries per slot, computed by fill_thresholds): (float) m > (float) COUNT * lowf  <=>  m >= ithr = floor((float) COUNT * lowf) + 1,
// and bucket 00 = COUNT - t reaches ithr  <=>  t <= jthr = COUNT - ithr.
template <int NW, int WS, int NWW>
__device__ __forceinline__ u64 filter_k_uni(const u32 (&P1)[NW], const u32 (&P2)[NW], const u32 (&P3)[NW], int L, int k,
                                            const int2 th, u64 hasm) {
    const u32 bs = (u32) k & 31u;
    const int W = L - k + 1;                  // COUNT, wave-uniform
    const int lastbits = W - 32 * (NWW - 1);  // windows in word NWW-1: [0, 32) by construction of FilterRangeUni
    const u32 wm = (1u << (lastbits & 31)) - 1u;
    u32 F1[NWW], F2[NWW];
    u32 c1x = 0, cx1 = 0, c11 = 0;
#pragma unroll
    for (int j = 0; j < NWW; j++) {
        F1[j] = P1[j] ^ shr_word<NW, WS>(P1, j, bs);
        F2[j] = P2[j] ^ shr_word<NW, WS>(P2, j, bs);
        if (j == NWW - 1) {
            F1[j] &= wm;
            F2[j] &= wm;
            // keep the masked words as values: otherwise the compiler re-derives them inside every
            // 3-input bit op below and spends two extra xors per k
            asm volatile("" : "+v"(F1[j]), "+v"(F2[j]));
        }
        // v_bcnt_u32_b32 adds its second operand: chaining the words' popcounts costs no separate add (left to itself
        // the compiler counts three words independently and spends a v_add3 per bucket)
        c1x = bcnt_acc(F1[j], c1x);
        cx1 = bcnt_acc(F2[j], cx1);
        c11 = bcnt_acc(F1[j] & F2[j], c11);
    }
    const u32 c10 = c1x - c11, c01 = cx1 - c11;
    const u32 m3 = max(max(c10, c01), c11);
    const int t = (int) (c1x + c01);  // COUNT - c00
    const u64 p4 = __ballot(m3 >= (u32) th.x) | __ballot(t <= th.y);  // two compares straight into SGPR masks
    u64 pm = p4;
    if (p4 & ~hasm) {  // third parity, as in filter_k: only for a lane's first candidate
        u32 c001 = 0, c010 = 0, c011 = 0, c100 = 0, c101 = 0, c110 = 0, c111 = 0;
#pragma unroll
        for (int j = 0; j < NWW; j++) {
            const u32 F3 = P3[j] ^ shr_word<NW, WS>(P3, j, bs);
            const u32 v = j == NWW - 1 ? wm : 0xffffffffu;
            const u32 a11 = F1[j] & F2[j], a10 = F1[j] & ~F2[j], a01 = ~F1[j] & F2[j], a00 = v & ~(F1[j] | F2[j]);
            const u32 b111 = a11 & F3, b101 = a10 & F3, b011 = a01 & F3, b001 = a00 & F3;
            c111 += __popc(b111);
            c110 += __popc(a11 ^ b111);
            c101 += __popc(b101);
            c100 += __popc(a10 ^ b101);
            c011 += __popc(b011);
            c010 += __popc(a01 ^ b011);
            c001 += __popc(b001);
        }
        const u32 c000 = (u32) W - c001 - c010 - c011 - c100 - c101 - c110 - c111;
        const u32 m8 = max(max(max(c000, c001), max(c010, c011)), max(max(c100, c101), max(c110, c111)));
        const u64 p8 = __ballot(m8 >= (u32) th.x);
        pm = (p4 & hasm) | (p8 & ~hasm);
    }
    return pm;
}

// what one segment's k loop accumulates
struct UniVerdict {
    u64 hasm;   // wave mask: lanes with a candidate at any k of the loop so far
    u64 flagm;  // wave mask: lanes with a candidate inside the segment's own [kmin, kmax]
    u32 clo, chi;  // this lane's candidate mask (only maintained when `dbg`)
};

// as FilterRange, with the word count taken from the segment's own (uniform) length.
// SLOW: some k of the loop lie outside the segment's [kmin, kmax], or the per-lane masks are wanted
// (trew_hip_filter_masks); otherwise flagm is simply hasm after the loop and the per-k bookkeeping
// is one scalar OR.  The thresholds of the next k are fetched from LDS one iteration ahead.
template <int NW, int WS, int NWW, bool SLOW>
struct FilterRangeUni {
    static __device__ __forceinline__ void run(const u32 (&P1)[NW], const u32 (&P2)[NW], const u32 (&P3)[NW], int klo, int khi, int L,
                                               int kmin, int kmax, const int2 *__restrict__ thr_row, bool dbg, UniVerdict &vd) {
        int a = L + 2 - 32 * NWW, b = L + 1 - 32 * (NWW - 1);
        if (NWW == NW) a = klo;
        a = a < klo ? klo : a;
        b = b > khi ? khi : b;
        if (a <= b) {
            int2 th_next = thr_row[a - 1];
            for (int k = a; k <= b; k++) {
                const int2 th = th_next;
                th_next = thr_row[k];  // entry 64 exists (padding)
                const u64 pm = filter_k_uni<NW, WS, NWW>(P1, P2, P3, L, k, th, vd.hasm);
                vd.hasm |= pm;
                if (SLOW) {
                    if (k >= kmin && k <= kmax) vd.flagm |= pm;
                    if (dbg) {  // wave-uniform
                        const bool mine = (pm >> lane_id()) & 1ull;
                        if (k <= 32)
                            vd.clo |= mine ? (1u << ((k - 1) & 31)) : 0u;
                        else
                            vd.chi |= mine ? (1u << ((k - 33) & 31)) : 0u;
                    }
                }
            }
        }
        FilterRangeUni<NW, WS, NWW - 1, SLOW>::run(P1, P2, P3, klo, khi, L, kmin, kmax, thr_row, dbg, vd);
    }
};
template <int NW, int WS, bool SLOW>
struct FilterRangeUni<NW, WS, 0, SLOW> {
    static __device__ __forceinline__ void run(const u32 (&)[NW], const u32 (&)[NW], const u32 (&)[NW], int, int, int, int, int, const int2 *,
                                               bool, UniVerdict &) {}
};

// N-free segment of wave-uniform length L (same verdicts as filter_segment): vd.flagm = lanes with a
// candidate k, vd.clo/chi = the lane's candidate mask when dbg
template <int NW>
__device__ __forceinline__ void filter_segment_uni(const u32 (&lo)[NW], const u32 (&hi)[NW], int L, int kmin, int kmax, int gmin, int gmax,
                                                   const int2 *__restrict__ thr_row, bool dbg, UniVerdict &vd) {
    u32 P1[NW], P2[NW], P3[NW];
    {
        u32 f1[NW], f2[NW], f3[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int bits = L - 32 * j;
            const u32 lm = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
            f1[j] = lo[j] & lm;
            f2[j] = hi[j] & lm;
            f3[j] = f1[j] & f2[j];
        }
        prefix_parity<NW>(f1, P1);
        prefix_parity<NW>(f2, P2);
        prefix_parity<NW>(f3, P3);
    }
    vd.hasm = vd.flagm = 0;
    vd.clo = vd.chi = 0;
    const int g31 = gmax < 31 ? gmax : 31;
    if (dbg || kmin > gmin || kmax < gmax) {  // wave-uniform
        FilterRangeUni<NW, 0, NW, true>::run(P1, P2, P3, gmin, g31, L, kmin, kmax, thr_row, dbg, vd);
        FilterRangeUni<NW, 1, NW, true>::run(P1, P2, P3, gmin > 32 ? gmin : 32, gmax < 63 ? gmax : 63, L, kmin, kmax, thr_row, dbg, vd);
        FilterRangeUni<NW, 2, NW, true>::run(P1, P2, P3, gmin > 64 ? gmin : 64, gmax, L, kmin, kmax, thr_row, dbg, vd);
    } else {
        FilterRangeUni<NW, 0, NW, false>::run(P1, P2, P3, gmin, g31, L, kmin, kmax, thr_row, false, vd);
        FilterRangeUni<NW, 1, NW, false>::run(P1, P2, P3, gmin > 32 ? gmin : 32, gmax < 63 ? gmax : 63, L, kmin, kmax, thr_row, false, vd);
        FilterRangeUni<NW, 2, NW, false>::run(P1, P2, P3, gmin > 64 ? gmin : 64, gmax, L, kmin, kmax, thr_row, false, vd);
        vd.flagm = vd.hasm;
    }
}

// Block size of the prefilter.  Measured on MI355X (tools/filter_grid_ab.sh, profiles/r02/README.md): 64-thread blocks
// (no block-level barrier at all) and 256-thread blocks run the same 0.70-0.72 ms on 10 M reads, and a variant with
// wave-private staging lists was slower (95 VGPRs, 5 waves per SIMD) -- the kernel is bound by VALU issue, not by
// its barriers.  Twice as many blocks as are resident is worth 5 % (the dispatcher back-fills the uneven tail).
constexpr u32 kFilterThreads = TREW_FILTER_THREADS;
constexpr u32 kStage = kFilterThreads == 256 ? 1024 : 192;  // unit indices a block stages in LDS before one global append
constexpr u32 kDefer = 2 * kFilterThreads;                   // units set aside by the fast path (drained one block-full at a time)

// Persistent blocks, grid-stride over the reads.  Survivors are staged in LDS and appended to
// the worklist with ONE global atomic per flush: a per-wave atomic on the single worklist
// counter caps at ~88 appends/us on MI355X (MI355X_MICROARCH.md "dequeue"), which was as long
// as the whole k loop.
template <int NW>
__global__ __launch_bounds__(kFilterThreads) void filter_kernel(DevParams P, DevBatch B, u32 *wl, u32 *wl_count, u32 wl_cap,
                                                                u64 *dbg_masks, int dbg_slots, int max_seg, u32 *diag,
                                                                const int2 *__restrict__ thr_tab) {
    __shared__ u32 stage[kStage];
    __shared__ u32 stage_n, flush_base;
    if (threadIdx.x == 0) stage_n = 0;
    __syncthreads();
    auto flush = [&]() __attribute__((always_inline)) {  // block-uniform
        const u32 n = stage_n;
        if (n) {
            if (threadIdx.x == 0) flush_base = atomicAdd(wl_count, n);
            __syncthreads();
            const u32 fb = flush_base;
            for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
                if (fb + i < wl_cap)
                    wl[fb + i] = stage[i];
                else
                    atomicAdd(&diag[kDiagWorklistDrop], 1u);  // surfaced by trew_hip_collect: never silent
            }
            __syncthreads();
            if (threadIdx.x == 0) stage_n = 0;
        }
        __syncthreads();
    };
    // wave-aggregated append of the flagged units to the block's LDS stage (block-uniform call)
    auto append = [&](bool flag, u32 unit) __attribute__((always_inline)) {
        const u64 bal = __ballot(flag);
        if (bal) {
            const u32 lane = lane_id();
            const int leader = __ffsll((long long) bal) - 1;
            u32 sb = 0;
            if ((int) lane == leader) sb = atomicAdd(&stage_n, (u32) __popcll(bal));
            sb = __shfl(sb, leader);
            if (flag) stage[sb + (u32) __popcll(bal & ((1ull << lane) - 1ull))] = unit;  // < kStage: flushed below when > kStage - block size
        }
        __syncthreads();
        if (stage_n > kStage - kFilterThreads) flush();
    };
    const int nslots = mode_slots(P.mode);
    const int gmax_run = (P.flags & TREW_FLAG_DEBUG_NO_KLOOP) ? P.min_mer - 1 : P.max_mer;
    // general path: any read length, N anywhere
    auto general = [&](u64 unit, bool active) __attribute__((always_inline)) -> u64 {
        ReadRef rd[2];
        rd[0].w = B.words;
        rd[0].len = 0;
        rd[0].nw = 0;
        rd[1] = rd[0];
        if (active) {
            if (P.mode == TREW_MODE_PAIR) {
                rd[0] = get_read(B, 2 * unit);
                rd[1] = get_read(B, 2 * unit + 1);
            } else {
                rd[0] = get_read(B, unit);
            }
        }
        u64 any = 0;
#pragma unroll
        for (int slot = 0; slot < kMaxSlots; slot++) {
            if (slot < nslots) {
                u64 mask = 0;
                Segment sg = get_segment(P.mode, slot, rd[0].len, rd[1].len, P.min_mer, P.max_mer, P.slice_len);
                const bool ok = active && sg.valid && sg.len <= (u32) (32 * NW - 1);
                if (__any(ok)) {
                    u32 lo[NW], hi[NW], nm[NW];
                    const ReadRef &r = sg.mate ? rd[1] : rd[0];
                    if (ok) {
                        load_planes<NW>(r, sg.start, lo, hi, nm);
                    } else {
#pragma unroll
                        for (int j = 0; j < NW; j++) {
                            lo[j] = hi[j] = 0;
                            nm[j] = 0xffffffffu;
                        }
                    }
                    u64 m;
                    if (P.flags & TREW_FLAG_NO_FILTER)
                        m = all_k_mask(sg.kmin, sg.kmax);
                    else
                        m = filter_segment<NW>(lo, hi, nm, ok ? (int) sg.len : 0, sg.kmin, sg.kmax, P.min_mer, gmax_run, max_seg, P.lowf);
                    mask = ok ? m : 0ull;
                }
                // a segment too long for this instantiation must never be dropped silently
                if (active && sg.valid && sg.len > (u32) (32 * NW - 1)) mask = all_k_mask(sg.kmin, sg.kmax);
                any |= mask;
                if (dbg_masks && active && slot < dbg_slots) dbg_masks[unit * (u64) dbg_slots + slot] = mask;
            }
        }
        return any;
    };

    // One loop for both kinds of batch, so that the (large) general path is instantiated once:
    // every round the block takes one fresh unit per thread; the fast path judges the N-free ones on the spot
    // and sets the others aside, a batch without uniform geometry sets all of them aside; whenever
    // a block-full of units is waiting (or the input is exhausted) the general path drains them.
    __shared__ u32 defer[kDefer];
    __shared__ u32 defer_n;
    // thr_tab[slot * kThrRow + k - 1] = pass thresholds of (slot, k) for this batch's uniform geometry (fill_thresholds, below): read-only
    // global memory at a wave-uniform address, i.e. scalar loads into SGPRs -- the k loops spend no vector instruction on them
    const u32 UL = B.uniform_length;
    const bool uni = NW <= 5 && UL != 0 && thr_tab != nullptr && P.mode != TREW_MODE_LONG && !(P.flags & TREW_FLAG_NO_FILTER);
    if (threadIdx.x == 0) defer_n = 0;
    __syncthreads();
    u64 base = (u64) blockIdx.x * blockDim.x;
    for (;;) {
        const bool more = base < B.n_units;  // block-uniform
        if (more) {
            const u64 unit = base + threadIdx.x;
            const bool active = unit < B.n_units;
            base += (u64) gridDim.x * blockDim.x;
            u64 any = 0;
            bool dfr = active;
            if constexpr (NW <= 5) {
                if (uni) {
                    ReadRef rd[2];
                    rd[0].w = B.words;
                    rd[0].len = 0;
                    rd[0].nw = 0;
                    rd[1] = rd[0];
                    if (active) {
                        if (P.mode == TREW_MODE_PAIR) {
                            rd[0] = get_read(B, 2 * unit);
                            rd[1] = get_read(B, 2 * unit + 1);
                        } else {
                            rd[0] = get_read(B, unit);
                        }
                    }
                    dfr = false;
                    for (int slot = 0; slot < nslots; slot++) {  // wave-uniform geometry: no need to unroll
                        const Segment sg = get_segment(P.mode, slot, UL, UL, P.min_mer, P.max_mer, P.slice_len);
                        if (!sg.valid) continue;
                        if (sg.len > (u32) (32 * NW - 1)) {  // too long for this instantiation: the general path keeps every k
                            dfr = dfr || active;
                            continue;
                        }
                        u32 lo[NW], hi[NW], nm[NW];
                        const ReadRef r = sg.mate ? rd[1] : rd[0];
                        load_planes<NW>(r, sg.start, lo, hi, nm);
                        u32 anyn = 0;
#pragma unroll
                        for (int j = 0; j < NW; j++) {
                            const int bits = (int) sg.len - 32 * j;
                            const u32 lm = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
                            anyn |= nm[j] & lm;
                        }
                        UniVerdict vd;
                        filter_segment_uni<NW>(lo, hi, (int) sg.len, sg.kmin, sg.kmax, P.min_mer, gmax_run, thr_tab + slot * kThrRow, dbg_masks != nullptr, vd);
                        dfr = dfr || (active && anyn != 0);
                        any |= (vd.flagm >> lane_id()) & 1ull;
                        if (dbg_masks && active && anyn == 0 && slot < dbg_slots)
                            dbg_masks[unit * (u64) dbg_slots + slot] = ((((u64) vd.chi) << 32) | vd.clo) & all_k_mask(sg.kmin, sg.kmax);
                    }
                }
            }
            {
                const u64 bal = __ballot(dfr);
                if (bal) {
                    const u32 lane = lane_id();
                    const int leader = __ffsll((long long) bal) - 1;
                    u32 sb = 0;
                    if ((int) lane == leader) sb = atomicAdd(&defer_n, (u32) __popcll(bal));
                    sb = __shfl(sb, leader);
                    if (dfr) defer[sb + (u32) __popcll(bal & ((1ull << lane) - 1ull))] = (u32) unit;  // < kDefer: drained below at one block-full
                }
            }
            append(active && !dfr && any != 0, (u32) unit);  // syncs the block
        }
        const u32 dn = defer_n;  // block-uniform: every add happened before the last barrier
        if (dn >= kFilterThreads || (!more && dn > 0u)) {
            const u32 take = dn < kFilterThreads ? dn : kFilterThreads;
            const u32 at = dn - take;
            const bool active2 = threadIdx.x < take;
            const u32 unit2 = active2 ? defer[at + threadIdx.x] : 0u;
            __syncthreads();
            if (threadIdx.x == 0) defer_n = at;
            const u64 any2 = general(unit2, active2);
            append(active2 && any2 != 0, unit2);
        } else if (!more) {
            break;
        }
    }
    flush();
}

// ------------------------------------------------------------------ count table
// Key layout (64 bits, one CAS claims a slot, lock-free and exact):
//   bit 63 valid | bits 62..60 table | bits 59..55 k-1 | bits 54..0 word >> 9
// The 9 low bits of the word select one of 512 partitions, so the full 2k-bit
// word (up to 64 bits at k = 32) is recoverable as (stored << 9) | partition.
__device__ __forceinline__ u64 hash64(u64 x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// the row's home (partition / wide table) is full: append it to the spill log
__attribute__((noinline)) __device__ void table_spill(DevTable T, int table, int k, u64 lo, u64 hi, u64 cnt) {
    const DevWide W = *T.wide;
    const u32 at = atomicAdd(W.spill_n, 1u);
    if (at < W.spill_cap) {
        trew_hip_row r;
        r.k = k;
        r.table = table;
        r.word_lo = lo;
        r.word_hi = hi;
        r.count = cnt;
        W.spill_rows[at] = r;
    } else {
        atomicExch(&T.overflow[kDiagOverflow], 1u);
    }
}

__attribute__((noinline)) __device__ void table_add(DevTable T, int table, int k, u64 word, u64 cnt) {
    if (T.log2_part_slots == 0xffffffffu) return;  // TREW_FLAG_DEBUG_NO_EMIT (timing experiments only)
    const u32 part = (u32) (word & ((1u << kTablePartBits) - 1u));
    const u64 key = (1ull << 63) | ((u64) table << 60) | ((u64) (k - 1) << 55) | (word >> kTablePartBits);
    const u32 S = 1u << T.log2_part_slots, mask = S - 1u;
    const u32 h = (u32) hash64(key) & mask;
    const u64 base = (u64) part << T.log2_part_slots;
    for (u32 probe = 0; probe < S; probe++) {
        const u64 idx = base + ((h + probe) & mask);
        u64 cur = __hip_atomic_load(&T.keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            u64 expected = 0;
            if (__hip_atomic_compare_exchange_strong(&T.keys[idx], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                cur = key;
                atomicAdd(&T.overflow[kDiagInserted], 1u);  // occupancy, read by trew_hip_table_pressure (new keys are rare)
            } else {
                cur = expected;
            }
        }
        if (cur == key) {
            __hip_atomic_fetch_add(&T.counts[idx], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    table_spill(T, table, k, word, 0ull, cnt);
}

// Wide entries (k in (32, 64], 128-bit words).  Slot = {tag, word_lo, word_hi, count};
// tag = bit 63 occupied | bit 62 ready | bits 61..59 table | bits 58..52 k | bits 51..0 hash(word).
// A slot is claimed by one CAS on the tag, its word is published, then the ready bit is set.
// Every access to a slot goes through device-scope atomic RMWs (the coherence point across the
// 8 XCD L2s), so no fence protocol is needed.  If a racing inserter ever fails to recognise its
// key it only creates a duplicate slot; trew_hip_collect merges duplicates (counts are sums).
__device__ __forceinline__ u64 coherent_load(u64 *p) { return atomicOr((unsigned long long *) p, 0ull); }

__attribute__((noinline)) __device__ void table_add_wide(DevTable T0, int table, int k, u64 lo, u64 hi, u64 cnt) {
    if (T0.log2_part_slots == 0xffffffffu) return;  // TREW_FLAG_DEBUG_NO_EMIT
    const DevWide T = *T0.wide;
    const u64 READY = 1ull << 62;
    const u64 h = hash64(lo ^ hash64(hi + (u64) k)) & ((1ull << 52) - 1ull);
    const u64 base = (1ull << 63) | ((u64) table << 59) | ((u64) k << 52) | h;
    const u32 S = 1u << T.wide_log2_slots, mask = S - 1u;
    u32 idx = (u32) hash64(base) & mask;
    for (u32 probe = 0; probe < S; probe++, idx = (idx + 1u) & mask) {
        u64 t = coherent_load(&T.wtag[idx]);
        if (t == 0) {
            const u64 prev = atomicCAS((unsigned long long *) &T.wtag[idx], 0ull, base);
            if (prev == 0) {
                atomicAdd(&T0.overflow[kDiagInsertedWide], 1u);
                atomicExch((unsigned long long *) &T.wlo[idx], lo);
                atomicExch((unsigned long long *) &T.whi[idx], hi);
                __threadfence();
                atomicOr((unsigned long long *) &T.wtag[idx], READY);
                atomicAdd((unsigned long long *) &T.wcount[idx], cnt);
                return;
            }
            t = prev;
        }
        if ((t & ~READY) == base) {
            for (int spin = 0; !(t & READY) && spin < (1 << 20); spin++) t = coherent_load(&T.wtag[idx]);
            if ((t & READY) && coherent_load(&T.wlo[idx]) == lo && coherent_load(&T.whi[idx]) == hi) {
                atomicAdd((unsigned long long *) &T.wcount[idx], cnt);
                return;
            }
        }
    }
    table_spill(T0, table, k, lo, hi, cnt);
}

__device__ __forceinline__ void table_add(DevTable T, int table, int k, u128 word, u64 cnt) {
    if (k <= 32)
        table_add(T, table, k, (u64) word, cnt);
    else
        table_add_wide(T, table, k, (u64) word, (u64) (word >> 64), cnt);
}

// ------------------------------------------------------------------ exact path
// LDS working set of one wave, carved from dynamic LDS and sized by the longest
// segment of the batch (cap bases, a multiple of 64): 2.6 KB for 150-bp reads, so
// 8 waves per SIMD stay resident and hide the global/LDS latency chains.
// The struct only carries the two sizes (it travels in SGPRs); every array is an
// offset from the dynamic-LDS base so that accesses compile to ds_* instructions.
//   seq   [2][cap/32+2] u64  2-bit bases, first base most significant (KmerSeq orientation, kmer.h:77)
//   vmask [cap/64+2] u64  bit i: window i has no N
//   emask [cap/64+2] u64  bit i: base i == base i+k (Lemma A: windows i, i+1 share a class)
//   nmask [2][cap/32+2] u32  bit i = base i is not A/C/G/T or lies past the end of the staged range
// seq/nmask hold a whole read (two of them: both mates of a pair), staged ONCE per read; a segment
// is a view: s0 = its first base inside the staged bases.  Windows never reach past the segment
// (i < L-k+1), so the view needs no end mask of its own.  Long reads do not fit: the long driver
// stages one slice at a time at base 0.
//   raw   [4][rawwords] u32  packed triples of the chunk's reads (or of the two mates), staged once
//   cnt   [cap] u16  class size at the class's first item, else 0
//   start [cap] u16  first window of each run
//   intent [32] u32  per-chunk read descriptors of the short/segment driver (`meta`)
//   ckey/cpart/ccnt [kCacheSlots]  the wave's private count cache (see cached_add)
//   canon [cap]  WT   per run (fast path) or per window (fallback); WT = u64 (k <= 32) or u128
struct ExactSmem {
    u32 cap, rawwords;
    u32 s0;         // first base of the current segment within the staged bases
    u32 rs, rnw;    // the same segment in the read's packed triples: first base, triples of the read
    const u32 *rw;  // the read's triples (LDS copy or global), nullptr when unknown
};
__host__ __device__ inline u32 exact_rangewords(u32 cap) { return cap / 32 + 2; }  // words of one staged range (+ read-ahead)

constexpr u32 kCacheSlots = 128;
__host__ __device__ inline u32 exact_lds_precache(u32 cap, u32 rawwords) {  // everything before the count cache, 16-byte aligned
    const u32 b = 2 * exact_rangewords(cap) * 8 + 2 * (cap / 64 + 2) * 8 + 2 * exact_rangewords(cap) * 4 + 4 * rawwords * 4 + 2 * cap * 2 + 32 * 4;
    return (b + 15u) & ~15u;
}
__host__ __device__ inline u32 exact_lds_fixed(u32 cap, u32 rawwords) {  // everything before canon[], 16-byte aligned
    return exact_lds_precache(cap, rawwords) + kCacheSlots * 16u;
}
constexpr u32 kSaveItems = 16, kSaveSlots = 6;  // pair driver: class tables kept per slot between decide and flush
__host__ __device__ inline u32 exact_lds_bytes(u32 cap, u32 rawwords, u32 wordbytes) {
    return exact_lds_fixed(cap, rawwords) + cap * wordbytes + 16 + kSaveSlots * kSaveItems * (wordbytes + 2u);
}

__device__ __forceinline__ unsigned char *lds0() {
    extern __shared__ __attribute__((aligned(16))) unsigned char trew_lds[];
    return trew_lds;
}
// Phase profile (tools/phase_profile.py, built with -DTREW_PHASE_PROFILE only): lane 0 of each wave
// accumulates s_memtime deltas per phase in LDS and adds them to g_phase when the wave retires.
#ifdef TREW_PHASE_PROFILE
__device__ unsigned long long g_phase[32];
__device__ __forceinline__ unsigned long long *ph_lds() {
    __shared__ unsigned long long ph[32];
    return ph;
}
#define PH_T0(v) const unsigned long long v = (unsigned long long) clock64()
#define PH_ADD(i, v)                                                                  \
    do {                                                                              \
        if (lane_id() == 0) ph_lds()[i] += (unsigned long long) clock64() - (v);      \
    } while (0)
#define PH_CNT(i, c)                                   \
    do {                                               \
        if (lane_id() == 0) ph_lds()[i] += (c);        \
    } while (0)
#else
#define PH_T0(v)
#define PH_ADD(i, v)
#define PH_CNT(i, c)
#endif
enum { PH_TOTAL = 0, PH_STAGE, PH_LOADSEG, PH_BOUNDS, PH_DECIDE, PH_RUNS, PH_WINDOWS, PH_RECORD_EVAL, PH_EMIT, PH_FLUSH, PH_EVALK_A, PH_PAIR_STAGE, PH_PAIR_FLUSH, PH_PAIR_FWD, PH_PAIR_BWD, PH_PAIR_WHOLE,
       PH_N_READS = 16, PH_N_RUNS_CALLS, PH_N_WINDOWS_CALLS, PH_N_RECORD, PH_N_RUNS_TOTAL, PH_N_K5 };

__device__ __forceinline__ u64 *sm_seq(ExactSmem sm) { return (u64 *) lds0(); }
__device__ __forceinline__ u64 *sm_vmask(ExactSmem sm) { return sm_seq(sm) + 2 * exact_rangewords(sm.cap); }
__device__ __forceinline__ u64 *sm_emask(ExactSmem sm) { return sm_vmask(sm) + (sm.cap / 64 + 2); }
__device__ __forceinline__ u32 *sm_nmask(ExactSmem sm) { return (u32 *) (sm_emask(sm) + (sm.cap / 64 + 2)); }
__device__ __forceinline__ u32 *sm_raw(ExactSmem sm) { return sm_nmask(sm) + 2 * exact_rangewords(sm.cap); }
__device__ __forceinline__ unsigned short *sm_cnt(ExactSmem sm) { return (unsigned short *) (sm_raw(sm) + 4 * sm.rawwords); }
__device__ __forceinline__ unsigned short *sm_start(ExactSmem sm) { return sm_cnt(sm) + sm.cap; }
__device__ __forceinline__ u32 *sm_intent(ExactSmem sm) { return (u32 *) (sm_start(sm) + sm.cap); }
__device__ __forceinline__ u64 *sm_ckey(ExactSmem sm) { return (u64 *) (lds0() + exact_lds_precache(sm.cap, sm.rawwords)); }
__device__ __forceinline__ u32 *sm_cpart(ExactSmem sm) { return (u32 *) (sm_ckey(sm) + kCacheSlots); }
__device__ __forceinline__ u32 *sm_ccnt(ExactSmem sm) { return sm_cpart(sm) + kCacheSlots; }
template <typename WT>
__device__ __forceinline__ WT *sm_canon(ExactSmem sm) { return (WT *) (lds0() + exact_lds_fixed(sm.cap, sm.rawwords)); }
template <typename WT>
__device__ __forceinline__ WT *sm_save_canon(ExactSmem sm) { return sm_canon<WT>(sm) + sm.cap + 16 / sizeof(WT); }  // [kSaveSlots][kSaveItems]
template <typename WT>
__device__ __forceinline__ unsigned short *sm_save_cnt(ExactSmem sm) { return (unsigned short *) (sm_save_canon<WT>(sm) + kSaveSlots * kSaveItems); }

__device__ __forceinline__ ExactSmem uni(ExactSmem sm) {
    sm.cap = rfl(sm.cap);
    sm.rawwords = rfl(sm.rawwords);
    sm.s0 = rfl(sm.s0);
    sm.rs = rfl(sm.rs);
    sm.rnw = rfl(sm.rnw);
    sm.rw = rfl_ptr(sm.rw);
    return sm;
}
__device__ __forceinline__ DevTable uni(DevTable T) {
    T.keys = rfl_ptr(T.keys);
    T.counts = rfl_ptr(T.counts);
    T.log2_part_slots = rfl(T.log2_part_slots);
    T.overflow = rfl_ptr(T.overflow);
    T.wide = rfl_ptr(T.wide);
    return T;
}

__device__ __forceinline__ u64 spread32(u32 v) {
    u64 x = v;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// stage bases [s, s+L) of a read into LDS (one wave)
// copy a read's triples into LDS once; later segment staging reads LDS, not HBM
__device__ ReadRef stage_read(ExactSmem sm, const ReadRef &rd, int mate) {
    u32 *dst = sm_raw(sm) + (u32) mate * sm.rawwords;
    const u32 n = 3u * rd.nw;
    if (n > sm.rawwords) return rd;  // does not fit (long mode): keep reading global memory
    for (u32 j = lane_id(); j < n; j += 64) dst[j] = rd.w[j];
    ReadRef r = rd;
    r.w = dst;
    return r;
}

// stage bases [s, s+L) of a read as range `range` (0 or 1) of seq[] / nmask[] (one wave)
__attribute__((noinline)) __device__ void stage_bases(ExactSmem sm, ReadRef rd, u32 s, u32 L, u32 range) {
    PH_T0(t_ph);
    sm = uni(sm);
    rd.w = rfl_ptr(rd.w);
    rd.len = rfl(rd.len);
    rd.nw = rfl(rd.nw);
    s = rfl(s);
    L = rfl(L);
    range = rfl(range);
    __syncthreads();
    const u32 nwords = (L + 31u) >> 5;
    const u32 segwords = exact_rangewords(sm.cap);
    u64 *seq = sm_seq(sm) + range * segwords;
    u32 *nmk = sm_nmask(sm) + range * segwords;
    for (u32 lane = lane_id(); lane < segwords; lane += 64) {
        u64 sq = 0;
        u32 nmv = 0xffffffffu;
        if (lane < nwords) {
            u32 lo[1], hi[1], nm[1];
            load_planes<1>(rd, s + 32u * lane, lo, hi, nm);
            const int bits = (int) L - 32 * (int) lane;
            const u32 lm = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
            nmv = nm[0] | ~lm;
            const u32 l = lo[0] & ~nmv, h = hi[0] & ~nmv;
            sq = spread32(__brev(l)) | (spread32(__brev(h)) << 1);
        }
        seq[lane] = sq;
        nmk[lane] = nmv;
    }
    __syncthreads();
    PH_ADD(PH_LOADSEG, t_ph);
}
// first base of staged range r
__device__ __forceinline__ u32 range_base(ExactSmem sm, u32 r) { return r * 32u * exact_rangewords(sm.cap); }
// the segment [s, s+L) of a read whose bases are staged as range r: a view, nothing moves
__device__ __forceinline__ ExactSmem view_segment(ExactSmem sm, u32 r, u32 s, const ReadRef &rd) {
    sm.s0 = range_base(sm, r) + s;
    sm.rs = s;
    sm.rnw = rd.nw;
    sm.rw = rd.w;
    return sm;
}
// long mode: stage one slice at base 0 and look at it
__device__ __forceinline__ ExactSmem load_segment(ExactSmem sm, ReadRef rd, u32 s, u32 L) {
    stage_bases(sm, rd, s, L, 0);
    sm.s0 = 0;
    sm.rs = s;
    sm.rnw = rd.nw;
    sm.rw = rd.w;
    return sm;
}


template <typename WT>
__device__ __forceinline__ WT kmask(int k) {
    return 2 * k >= (int) (8 * sizeof(WT)) ? ~(WT) 0 : ((((WT) 1) << (2 * k)) - 1);
}
__device__ __forceinline__ u32 popc_word(u64 w) { return (u32) __popcll(w); }
__device__ __forceinline__ u32 popc_word(u128 w) { return (u32) __popcll((u64) w) + (u32) __popcll((u64) (w >> 64)); }

// value of lane `src` (wave-uniform index) for every lane
__device__ __forceinline__ u64 readlane_word(u64 v, int src) {
    const u32 lo = (u32) __builtin_amdgcn_readlane((int) (u32) v, src), hi = (u32) __builtin_amdgcn_readlane((int) (u32) (v >> 32), src);
    return ((u64) hi << 32) | lo;
}
__device__ __forceinline__ u128 readlane_word(u128 v, int src) {
    return ((u128) readlane_word((u64) (v >> 64), src) << 64) | readlane_word((u64) v, src);
}

// get_rot_seq / get_rot_seq_128, kmer.cpp:1815-1833
template <typename WT>
__device__ __forceinline__ WT min_rotation(WT w, int k) {
    const int sh = 2 * (k - 1);
    if (k <= 16) {  // wave-uniform: the 2k-bit word fits 32 bits
        // rotation i (right by i bases) = bits [2i, 2i+2k) of the word written twice
        const u32 w32 = (u32) w;
        const u64 dup = ((u64) w32 << (2 * k)) | w32;
        const u32 dlo = (u32) dup, dhi = (u32) (dup >> 32);
        const u32 km = 2 * k >= 32 ? 0xffffffffu : ((1u << (2 * k)) - 1u);
        u32 ans = w32;
        if (k <= 8) {  // the doubled word fits 32 bits: one bit-field extract per rotation
            for (int i = 1; i < k; i++) ans = min(ans, __builtin_amdgcn_ubfe(dlo, 2u * (u32) i, 2u * (u32) k));
        } else {
            for (int i = 1; i < k; i++) ans = min(ans, alignbit(dhi, dlo, 2u * (u32) i) & km);
        }
        return ans;
    }
    if (sizeof(WT) > 8 && k <= 32) {  // fits 64 bits
        u64 tmp = (u64) w, ans = (u64) w;
        for (int i = 0; i < k - 1; i++) {
            tmp = ((tmp & 3ull) << sh) | (tmp >> 2);
            ans = tmp < ans ? tmp : ans;
        }
        return ans;
    }
    WT tmp = w, ans = w;
    for (int i = 0; i < k - 1; i++) {
        tmp = ((tmp & 3) << sh) | (tmp >> 2);
        ans = tmp < ans ? tmp : ans;
    }
    return ans;
}
// reverse the 32 2-bit groups of a 64-bit word and complement them
// (bit reversal is one v_bfrev per half; it also swaps the two bits of every group, which one more step undoes)
__device__ __forceinline__ u64 revcomp_groups64(u64 x) {
    const u32 lo = __brev((u32) (x >> 32)), hi = __brev((u32) x);  // reversed halves, swapped
    const u32 l2 = ((lo >> 1) & 0x55555555u) | ((lo & 0x55555555u) << 1), h2 = ((hi >> 1) & 0x55555555u) | ((hi & 0x55555555u) << 1);
    return ~(((u64) h2 << 32) | l2);
}
// reverse_complement_64(x) >> 2*(32-k), kmer.cpp:47-54 / 1987
__device__ __forceinline__ u64 revcomp(u64 x, int k) { return revcomp_groups64(x) >> (2 * (32 - k)); }
// reverse_complement_128(x) >> 2*(64-k), kmer.cpp:62-70
__device__ __forceinline__ u128 revcomp(u128 x, int k) {
    const u128 r = ((u128) revcomp_groups64((u64) x) << 64) | revcomp_groups64((u64) (x >> 64));
    return r >> (2 * (64 - k));
}
// get_repeat_check, kmer.cpp:1835-1867: the word uses a single base
template <typename WT>
__device__ __forceinline__ bool is_homopolymer(WT w, int k) {
    WT fives = (WT) 0x5555555555555555ull;
    if (sizeof(WT) > 8) fives |= fives << 64 % (8 * sizeof(WT));
    return w == ((w & 3) * (fives & kmask<WT>(k)));
}

template <typename WT>
struct KStat {
    u32 count;   // K_MER_DATA_COUNT
    u32 maxc;    // K_MER_DATA_MAX
    WT maxseq;   // K_MER_DATA_MAX_SEQ
    u32 n_items; // entries of canon[]/cnt[] left in LDS for emit_k
    bool pruned; // the bucket bound proved MAX/COUNT < need: maxc/maxseq were not computed
};

__device__ __forceinline__ u64 rfl_word(u64 v) { return rfl64(v); }
__device__ __forceinline__ u128 rfl_word(u128 v) { return ((u128) rfl64((u64) (v >> 64)) << 64) | rfl64((u64) v); }
// A KStat / Decision returned by a noinline function arrives in VGPRs; every field is wave-uniform.
// Saying so keeps the drivers' control state (chain counters, thresholds, intent lists) in SGPRs.
template <typename WT>
__device__ __forceinline__ KStat<WT> uni(KStat<WT> st) {
    st.count = rfl(st.count);
    st.maxc = rfl(st.maxc);
    st.maxseq = rfl_word(st.maxseq);
    st.n_items = rfl(st.n_items);
    st.pruned = rfl((u32) st.pruned) != 0;
    return st;
}

__device__ __forceinline__ u32 base_at(ExactSmem sm, u32 p) {
    p += sm.s0;
    return (u32) (sm_seq(sm)[p >> 5] >> (62u - 2u * (p & 31u))) & 3u;
}
template <typename WT>
__device__ __forceinline__ WT window_word(ExactSmem sm, u32 i, int k);
template <>
__device__ __forceinline__ u64 window_word<u64>(ExactSmem sm, u32 i, int k) {
    i += sm.s0;
    const u32 wi = i >> 5, sh = 2u * (i & 31u);
    const u64 a = sm_seq(sm)[wi], b = sm_seq(sm)[wi + 1];
    const u64 x = sh ? ((a << sh) | (b >> (64u - sh))) : a;
    return x >> (64 - 2 * k);
}
template <>
__device__ __forceinline__ u128 window_word<u128>(ExactSmem sm, u32 i, int k) {
    i += sm.s0;
    const u32 wi = i >> 5, sh = 2u * (i & 31u);
    const u64 a = sm_seq(sm)[wi], b = sm_seq(sm)[wi + 1], c = sm_seq(sm)[wi + 2];
    const u128 ab = ((u128) a << 64) | b;
    const u128 x = sh ? ((ab << sh) | (u128) (c >> (64u - sh))) : ab;
    return x >> (128 - 2 * k);
}
__device__ __forceinline__ bool window_valid(ExactSmem sm, u32 i, int k) {  // k <= 64
    i += sm.s0;
    const u32 wi = i >> 5, bi = i & 31u;
    const u64 lo = ((u64) sm_nmask(sm)[wi + 1] << 32) | sm_nmask(sm)[wi];
    const u64 hi = sm_nmask(sm)[wi + 2];
    const u64 nmw = bi ? ((lo >> bi) | (hi << (64u - bi))) : lo;
    const u64 km = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    return (nmw & km) == 0;  // no N inside the window (kmer.cpp:2190)
}

// wave argmax over per-lane (key, seq): largest class, ties to the class whose
// last window is earliest = the first to reach the maximum in scan order
// (strict '<' at kmer.cpp:2202).  key = (class size << 16) | (0xffff - last window)
template <typename WT>
__device__ __forceinline__ void wave_best(u32 best, WT best_seq, KStat<WT> &st) {
    const u32 m = wave_max_u32(best);
    const u64 who = __ballot(best == m && m != 0);
    if (who) {
        const int src = __ffsll((long long) who) - 1;
        st.maxc = m >> 16;
        st.maxseq = readlane_word(best_seq, src);
    }
}

// Fallback for segments with more than 64 runs: one item per window, class sizes
// by an all-pairs LDS-broadcast compare.  vmask[] must hold the valid-window bits.
template <typename WT>
__attribute__((noinline)) __device__ KStat<WT> eval_k_windows(ExactSmem sm, int W, int k, u32 count) {
    KStat<WT> st;  // returned by value: a reference parameter of a noinline function lives in scratch memory
    PH_T0(t_ph);
    PH_CNT(PH_N_WINDOWS_CALLS, 1);
    sm = uni(sm);
    W = rfl_i(W);
    k = rfl_i(k);
    count = rfl(count);
    st.count = count;
    st.maxc = 0;
    st.maxseq = 0;
    st.pruned = false;
    const u32 lane = lane_id();
    const int rounds = (W + 63) >> 6;
    for (int r = 0; r < rounds; r++) {
        const int i = r * 64 + (int) lane;
        if (i < W) {
            const bool valid = (sm_vmask(sm)[r] >> lane) & 1ull;
            sm_canon<WT>(sm)[i] = valid ? min_rotation<WT>(window_word<WT>(sm, (u32) i, k), k) : (WT) 0;
            sm_cnt(sm)[i] = 0;
        }
    }
    __syncthreads();
    u32 best = 0;
    WT best_seq = 0;
    for (int r = 0; r < rounds; r++) {
        const int i = r * 64 + (int) lane;
        const bool mine = i < W && ((sm_vmask(sm)[r] >> lane) & 1ull);
        const WT my = mine ? sm_canon<WT>(sm)[i] : (WT) 0;
        u32 c = 0, last = 0;
        bool first = true;
        for (int jr = 0; jr < rounds; jr++) {
            u64 vm = rfl64(sm_vmask(sm)[jr]);
            while (vm) {
                const int jb = __ffsll((long long) vm) - 1;
                vm &= vm - 1;
                const int j = jr * 64 + jb;
                const bool eq = sm_canon<WT>(sm)[j] == my;  // LDS broadcast read
                c += eq ? 1u : 0u;
                last = eq ? (u32) j : last;
                first = first && !(eq && j < i);
            }
        }
        if (mine) {
            if (first) sm_cnt(sm)[i] = (unsigned short) c;
            const u32 key = (c << 16) | (0xffffu - last);
            if (key > best) {
                best = key;
                best_seq = my;
            }
        }
    }
    wave_best<WT>(best, best_seq, st);
    st.n_items = (u32) W;
    PH_ADD(PH_WINDOWS, t_ph);
    return st;
}

template <typename WT>
__device__ KStat<WT> eval_runs(ExactSmem sm, int W, int k);

// One k of the counting loop of k_mer_check / k_mer_target (kmer.cpp:2183-2216,
// 1936-1967) on the segment staged in sm.  Lemma A: two
// adjacent valid windows i, i+1 are in the same rotation class iff base i ==
// base i+k, so the windows fall into maximal RUNS and only one canonical
// rotation per run is needed; runs with equal canonical word are then merged.
// Leaves canon[] / cnt[] in LDS for emit_k.  All lanes must call it; the
// result is wave-uniform.
//
// need > 0 asks for an early exit: windows of one class share their base
// composition, so the largest of the 8 (#lo, #hi, #A mod 2) parity buckets
// bounds MAX from above; if even that bound gives a frequency below `need`
// (the smallest threshold this k still has to reach in decide()), the k cannot
// be accepted and the per-run canonicalisation is skipped (st.pruned).
template <typename WT>
__attribute__((noinline)) __device__ KStat<WT> eval_k(ExactSmem sm, int L, int k, double need) {
    KStat<WT> st;
    st.count = 0;
    st.maxc = 0;
    st.maxseq = 0;
    st.n_items = 0;
    st.pruned = false;
    sm = uni(sm);
    L = rfl_i(L);
    k = rfl_i(k);
    need = rfl_f64(need);
    const int W = L - k + 1;
    if (W <= 0) return st;
    PH_T0(t_ph);
    const u32 lane = lane_id();
    const int rounds = (W + 63) >> 6;
    WT m5 = (WT) 0x5555555555555555ull;
    if (sizeof(WT) > 8) m5 |= m5 << 64 % (8 * sizeof(WT));
    m5 &= kmask<WT>(k);
    u32 b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
    __syncthreads();  // previous users of the LDS arrays are done
    if (need == 0.0 && sm.rw != nullptr && k < 64) {
        // No pruning wanted (record / k_mer_target): the window masks of this one k come straight from
        // the packed planes, lane j building 32 windows at once -- E = bases i and i+k agree, V = the
        // L-k+1 windows of an N-free segment -- instead of a walk over the windows.  A segment with an
        // N keeps the general walk below.
        const u32 segw = ((u32) L + 31u) >> 5;
        ReadRef rr;
        rr.w = sm.rw;
        rr.nw = sm.rnw;
        rr.len = 0;
        u32 lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, nm[3] = {0, 0, 0};
        if (lane < segw) load_planes<3>(rr, sm.rs + 32u * lane, lo, hi, nm);
        const int lbits = L - 32 * (int) lane;
        const u32 lm = lbits >= 32 ? 0xffffffffu : (lbits <= 0 ? 0u : ((1u << lbits) - 1u));
        if (!__any((nm[0] & lm) != 0u)) {
            const bool big = k >= 32;
            const u32 ks = (u32) k & 31u;
            const u32 slo = alignbit(big ? lo[2] : lo[1], big ? lo[1] : lo[0], ks);
            const u32 shi = alignbit(big ? hi[2] : hi[1], big ? hi[1] : hi[0], ks);
            const u32 e = ~((lo[0] ^ slo) | (hi[0] ^ shi));
            const int wbits = W - 32 * (int) lane;
            const u32 v = wbits >= 32 ? 0xffffffffu : (wbits <= 0 ? 0u : ((1u << wbits) - 1u));
            if (lane < 2u * ((u32) rounds + 1u)) {  // including the terminating zero word
                ((u32 *) sm_vmask(sm))[lane] = v;
                ((u32 *) sm_emask(sm))[lane] = e & v;
            }
            __syncthreads();
            PH_ADD(PH_EVALK_A, t_ph);
            return eval_runs<WT>(sm, W, k);
        }
    }
    for (int r = 0; r < rounds; r++) {
        const u32 i = (u32) r * 64u + lane;
        bool valid = false, eq = false, p1 = false, p2 = false, p3 = false;
        if ((int) i < W) {
            valid = window_valid(sm, i, k);
            const WT w = window_word<WT>(sm, i, k);
            eq = (u32) (w >> (2 * k - 2)) == base_at(sm, i + (u32) k);
            p1 = popc_word(w & m5) & 1;
            p2 = popc_word((w >> 1) & m5) & 1;
            p3 = popc_word(w & (w >> 1) & m5) & 1;
        }
        const u64 bv = __ballot(valid), be = __ballot(eq);
        if (lane == 0) {
            sm_vmask(sm)[r] = bv;
            sm_emask(sm)[r] = be;
        }
        if (need > 0.0) {
            const u64 f1 = __ballot(p1), f2 = __ballot(p2), f3 = __ballot(p3);
            const u64 a1 = bv & f1, a0 = bv ^ a1;
            const u64 a11 = a1 & f2, a10 = a1 ^ a11, a01 = a0 & f2, a00 = a0 ^ a01;
            const u64 c111 = a11 & f3, c101 = a10 & f3, c011 = a01 & f3, c001 = a00 & f3;
            b7 += (u32) __popcll(c111);
            b6 += (u32) __popcll(a11 ^ c111);
            b5 += (u32) __popcll(c101);
            b4 += (u32) __popcll(a10 ^ c101);
            b3 += (u32) __popcll(c011);
            b2 += (u32) __popcll(a01 ^ c011);
            b1 += (u32) __popcll(c001);
            b0 += (u32) __popcll(a00 ^ c001);
        }
    }
    if (lane == 0) {
        sm_vmask(sm)[rounds] = 0;
        sm_emask(sm)[rounds] = 0;
    }
    __syncthreads();
    if (need > 0.0) {
        u32 cnt = 0;
        for (int r = 0; r < rounds; r++) cnt += (u32) __popcll(rfl64(sm_vmask(sm)[r]));
        const u32 mb = max(max(max(b0, b1), max(b2, b3)), max(max(b4, b5), max(b6, b7)));
        // MAX <= mb and IEEE division is monotone in the numerator, so MAX/COUNT <= mb/COUNT < need
        if (cnt == 0 || (double) rfl(mb) / (double) cnt < need) {
            st.count = cnt;
            st.pruned = true;
            __syncthreads();
            return st;
        }
    }
    PH_ADD(PH_EVALK_A, t_ph);
    return eval_runs<WT>(sm, W, k);
}

// Second half of eval_k: vmask[] / emask[] (+ one zero word) are in LDS, visible to the wave.
template <typename WT>
__attribute__((noinline)) __device__ KStat<WT> eval_runs(ExactSmem sm, int W, int k) {
    KStat<WT> st;
    st.maxc = 0;
    st.maxseq = 0;
    st.n_items = 0;
    st.pruned = false;
    sm = uni(sm);
    W = rfl_i(W);
    k = rfl_i(k);
    const u32 lane = lane_id();
    const int rounds = (W + 63) >> 6;
    PH_T0(t_ph);
    PH_CNT(PH_N_RUNS_CALLS, 1);
    PH_CNT(PH_N_K5, k == 5 ? 1 : 0);
    // run starts: valid_i && !(valid_{i-1} && eq_{i-1}); compacted into start[]
    u32 R = 0, count = 0;
    u64 carry = 0;
    for (int r = 0; r < rounds; r++) {
        const u64 vm = rfl64(sm_vmask(sm)[r]), em = rfl64(sm_emask(sm)[r]);
        const u64 ve = vm & em;
        const u64 startmask = vm & ~((ve << 1) | carry);
        carry = ve >> 63;
        count += (u32) __popcll(vm);
        if ((startmask >> lane) & 1ull) sm_start(sm)[R + (u32) __popcll(startmask & ((1ull << lane) - 1ull))] = (unsigned short) (r * 64 + (int) lane);
        R += (u32) __popcll(startmask);
    }
    st.count = count;
    __syncthreads();
    if (R > 64) {
        st = eval_k_windows<WT>(sm, W, k, count);
        __syncthreads();
        PH_ADD(PH_RUNS, t_ph);
        return st;
    }
    // one lane per run
    WT canon = ~(WT) 0;
    u32 len = 0, s = 0;
    if (lane < R) {
        s = sm_start(sm)[lane];
        // run length = 1 + number of consecutive j >= s with valid_j && eq_j && valid_{j+1}
        u32 j = s;
        len = 1;
        for (;;) {
            const u32 wi = j >> 6, bi = j & 63u;
            const u64 v0 = sm_vmask(sm)[wi], v1 = sm_vmask(sm)[wi + 1];
            const u64 cw = (v0 & sm_emask(sm)[wi] & ((v0 >> 1) | (v1 << 63))) >> bi;
            const u64 inv = ~cw;
            const u32 ones = inv ? (u32) (__ffsll((long long) inv) - 1) : 64u;
            len += ones;
            if (ones + bi < 64u) break;
            j += ones;
            if (j >= (u32) W) break;
        }
        canon = min_rotation<WT>(window_word<WT>(sm, s, k), k);
    }
    const u32 end = s + len - 1;
    u32 tot = 0, last = 0;
    bool first = false;
    // One iteration per CLASS, not per run: the lowest run without a class is its leader, one ballot
    // finds every member, a DPP sum adds their lengths; runs are in position order, so the member in
    // the highest lane ends last.  (A TTAGGG read probed at k = 5 has ~50 runs in ~8 classes.)
    const u64 act = R >= 64u ? ~0ull : ((1ull << R) - 1ull);
    u64 remaining = act;
    while (remaining) {
        const int rp = __ffsll((long long) remaining) - 1;
        const WT other = readlane_word(canon, rp);
        const u64 eqm = __ballot(other == canon) & act;
        remaining &= ~eqm;
        const bool eq = (eqm >> lane) & 1ull;
        const u32 total = wave_sum_u32(eq ? len : 0u);
        const u32 oend = (u32) __builtin_amdgcn_readlane((int) end, 63 - __clzll((long long) eqm));
        if (eq) {
            tot = total;
            last = oend;
            first = (int) lane == rp;
        }
    }
    u32 key = 0;
    if (lane < R) {
        key = (tot << 16) | (0xffffu - last);
        sm_canon<WT>(sm)[lane] = canon;
        sm_cnt(sm)[lane] = first ? (unsigned short) tot : (unsigned short) 0;
    }
    wave_best<WT>(key, canon, st);
    st.n_items = R;
    __syncthreads();
    PH_CNT(PH_N_RUNS_TOTAL, R);
    PH_ADD(PH_RUNS, t_ph);
    return st;
}

// Wave-private count cache in LDS.  A few keys (the dominant motif's classes) receive an add from
// almost every surviving read; as device atomics on one address they serialise chip-wide
// (0.4 ms of a 1.3 ms launch, measured with TREW_FLAG_DEBUG_NO_EMIT).  Each wave keeps the first
// kCacheSlots distinct keys it meets in LDS (insert-only, no eviction: the hot keys show up in the
// first reads), adds to them with LDS atomics and flushes once when it runs out of work.
// Entry = the 64-bit table key of table_add + the 9 partition bits of the word.
__device__ __forceinline__ void cached_add(ExactSmem sm, DevTable T, int table, int k, u64 word, u32 cnt) {
    if (T.log2_part_slots == 0xffffffffu) return;  // TREW_FLAG_DEBUG_NO_EMIT
    const u32 part = (u32) (word & ((1u << kTablePartBits) - 1u));
    const u64 gkey = (1ull << 63) | ((u64) table << 60) | ((u64) (k - 1) << 55) | (word >> kTablePartBits);
    const u32 slot = (u32) (hash64(gkey ^ ((u64) part << 40)) >> 20) & (kCacheSlots - 1u);
    u64 *ckey = sm_ckey(sm);
    u32 *cpart = sm_cpart(sm), *ccnt = sm_ccnt(sm);
    // Claim and publish in two separated phases.  A lane that wins the CAS stores the partition bits of its key;
    // a lane of the same wave that lost the CAS to an equal gkey then reads them.  The cache is wave-private (one
    // wave per workgroup) and LDS operations of a wave execute in program order, so the only hazard is the
    // compiler moving the load above the store: the workgroup-scope fence + wave barrier between the phases
    // forbid that, and both accesses are atomic, so there is no data race in the formal sense either.
    u64 ek = __hip_atomic_load(&ckey[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool won = false;
    if (ek == 0) {
        const u64 prev = atomicCAS((unsigned long long *) &ckey[slot], 0ull, gkey);
        won = prev == 0;
        ek = won ? gkey : prev;
    }
    if (won) __hip_atomic_store(&cpart[slot], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const u32 spart = __hip_atomic_load(&cpart[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ek == gkey && spart == part)
        atomicAdd(&ccnt[slot], cnt);
    else
        table_add(T, table, k, word, (u64) cnt);
}
__device__ __forceinline__ void cached_add(ExactSmem sm, DevTable T, int table, int k, u128 word, u32 cnt) {
    if (k <= 32)
        cached_add(sm, T, table, k, (u64) word, cnt);
    else
        table_add(T, table, k, word, (u64) cnt);
}
__device__ void cache_clear(ExactSmem sm) {
    for (u32 i = lane_id(); i < kCacheSlots; i += 64) {
        sm_ckey(sm)[i] = 0;
        sm_cpart(sm)[i] = 0;
        sm_ccnt(sm)[i] = 0;
    }
    __syncthreads();
}
__device__ void cache_flush(ExactSmem sm, DevTable T) {
    __syncthreads();
    for (u32 i = lane_id(); i < kCacheSlots; i += 64) {
        const u64 gkey = sm_ckey(sm)[i];
        if (gkey) {
            const u64 word = ((gkey & ((1ull << 55) - 1ull)) << kTablePartBits) | sm_cpart(sm)[i];
            table_add(T, (int) ((gkey >> 60) & 7ull), (int) ((gkey >> 55) & 31ull) + 1, word, (u64) sm_ccnt(sm)[i]);
        }
    }
}

// add every class of the k just evaluated to the tables in table_mask (bit t).
// strand_canon: key = MIN(w, rot(rc(w))) (k_mer_target, kmer.cpp:1979-1988) else the
// rotation-canonical word itself (k_mer_check, kmer.cpp:2264-2313).
template <typename WT>
__attribute__((noinline)) __device__ void emit_k(ExactSmem sm, DevTable T, u32 n_items, int k, u32 table_mask, bool strand_canon) {
    sm = uni(sm);
    T = uni(T);
    n_items = rfl(n_items);
    k = rfl_i(k);
    table_mask = rfl(table_mask);
    strand_canon = rfl((u32) strand_canon) != 0;
    const u32 lane = lane_id();
    PH_T0(t_ph);
    for (u32 base = 0; base < n_items; base += 64) {  // wave-uniform trip count: the de-duplication below is a wave operation
        const u32 i = base + lane;
        u32 c = i < n_items ? sm_cnt(sm)[i] : 0u;
        WT w = 0;
        if (c) {
            w = sm_canon<WT>(sm)[i];
            if (strand_canon) {
                const WT rc = min_rotation<WT>(revcomp(w, k), k);
                w = rc < w ? rc : w;
            }
        }
        if (sizeof(WT) > 8 && k > 32 && strand_canon) {
            // A class and its reverse-complement class reach this point with the SAME strand-canonical key, in
            // two lanes of one wave.  In the wide table the lane that loses the slot claim waits for the winner's
            // ready bit (table_add_wide) -- a sibling lane of a lock-step wave must never be that winner, so equal
            // keys are merged here first (classes are distinct before canonicalisation: at most two lanes per key).
            u64 rem = __ballot(c != 0);
            while (rem) {
                const int src = __ffsll((long long) rem) - 1;
                const WT o = readlane_word(w, src);
                const u64 eqm = __ballot(c != 0 && w == o);
                const bool eq = (eqm >> lane) & 1ull;
                const u32 tot = wave_sum_u32(eq ? c : 0u);
                if (eq) c = (int) lane == src ? tot : 0u;
                rem &= ~eqm;
            }
        }
        if (c)
            for (u32 tm = table_mask; tm; tm &= tm - 1) cached_add(sm, T, __ffs((int) tm) - 1, k, w, c);
    }
    PH_ADD(PH_EMIT, t_ph);
}

// per-lane variable right shift of a multiword mask by off in [0, 63]
template <int NW>
__device__ __forceinline__ u32 shr_var_word(const u32 (&x)[NW], int j, u32 off) {
    const bool big = off >= 32u;
    const u32 x0 = x[j], x1 = j + 1 < NW ? x[j + 1 < NW ? j + 1 : 0] : 0u, x2 = j + 2 < NW ? x[j + 2 < NW ? j + 2 : 0] : 0u;
    return alignbit(big ? x2 : x1, big ? x1 : x0, off & 31u);
}

// Upper bound of MAX/COUNT for EVERY k of one segment at once: lane l handles
// k = gmin + l with the same bit-parallel parity-bucket bound as the prefilter
// (filter_segment), the shift amounts simply differ per lane.  Used by decide()
// to discard a candidate k with one readlane instead of a pass over its windows.
// Returns (double) maxbucket / (double) count, 0 where there is no valid window.
// what lane l knows about k = gmin + l of one segment
template <int NW>
struct LaneMasks {
    u32 V[NW];  // bit i: window i has no N and fits the segment
    u32 E[NW];  // bit i: base i == base i+k (meaningful where windows i and i+1 are both valid)
    double ub;  // maxbucket / COUNT, an upper bound of MAX / COUNT (0 where COUNT == 0)
    u32 runs;   // number of runs of adjacent same-class windows (Lemma A): what counting the classes of this k costs
};

template <int NW>
__device__ __forceinline__ void lane_bounds_at(const ReadRef &rd, u32 s, int L, int k, int gmax, LaneMasks<NW> &out);
// lane l <-> k = gmin + l of the segment [s, s+L)
template <int NW>
__device__ __forceinline__ void lane_bounds(const ReadRef &rd, u32 s, int L, int gmin, int gmax, LaneMasks<NW> &out) {
    lane_bounds_at<NW>(rd, s, L, gmin + (int) lane_id(), gmax, out);
}
// Two segments of one read in one pass when the k range fits half a wave (short reads: both
// halves): lanes 0..31 take segment A, lanes 32..63 segment B, k = gmin + (lane & 31).
template <int NW>
__device__ __forceinline__ void lane_bounds_pair(const ReadRef &rd, u32 sA, int LA, u32 sB, int LB, int gmin, int gmax, LaneMasks<NW> &out) {
    const bool hiHalf = lane_id() >= 32u;
    lane_bounds_at<NW>(rd, hiHalf ? sB : sA, hiHalf ? LB : LA, gmin + (int) (lane_id() & 31u), gmax, out);
}
// s, L, k may differ per lane (they do in lane_bounds_pair); everything below is per-lane arithmetic
template <int NW>
__device__ __forceinline__ void lane_bounds_at(const ReadRef &rd, u32 s, int L, int k, int gmax, LaneMasks<NW> &out) {
    PH_T0(t_ph);
    u32 lo[NW], hi[NW], nm[NW];
    load_planes<NW>(rd, s, lo, hi, nm);  // the lanes of one segment read the same (LDS-staged) words
    u32 v1[NW], P1[NW], P2[NW], P3[NW];
    {
        u32 f1[NW], f2[NW], f3[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
            int bits = L - 32 * j;
            u32 lm = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
            v1[j] = ~nm[j] & lm;
            f1[j] = lo[j] & v1[j];
            f2[j] = hi[j] & v1[j];
            f3[j] = f1[j] & f2[j];
        }
        prefix_parity<NW>(f1, P1);
        prefix_parity<NW>(f2, P2);
        prefix_parity<NW>(f3, P3);
    }
    const u32 ku = (u32) k;
    u32 V[NW];
    u32 anyn = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        int bits = L - 32 * j;
        u32 lm = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
        anyn |= nm[j] & lm;
    }
    if (anyn == 0) {
        // no N in the segment (wave-uniform: every lane sees the same segment): V_k = the L-k+1 lowest bits
        const int wbits = L - k + 1;
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int bits = wbits - 32 * j;
            V[j] = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
        }
    } else {
        // V_k[i] = AND_{t<k} v1[i+t] by binary decomposition of k over A_b = AND of b consecutive bases
        u32 A[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
            V[j] = 0xffffffffu;
            A[j] = v1[j];
        }
        u32 off = 0;
#pragma unroll
        for (int b = 1; b <= 64; b <<= 1) {
            if (b > 1) {  // A_b = A_{b/2} & (A_{b/2} >> b/2)
                u32 T2[NW];
#pragma unroll
                for (int j = 0; j < NW; j++) T2[j] = A[j] & shr_var_word<NW>(A, j, (u32) (b / 2));
#pragma unroll
                for (int j = 0; j < NW; j++) A[j] = T2[j];
            }
            if (b <= gmax) {  // wave-uniform
                const bool take = (ku & (u32) b) != 0;
#pragma unroll
                for (int j = 0; j < NW; j++) {
                    const u32 sh = shr_var_word<NW>(A, j, off);
                    V[j] &= take ? sh : 0xffffffffu;
                }
                off += take ? (u32) b : 0u;
            }
        }
    }
    u32 c000 = 0, c001 = 0, c010 = 0, c011 = 0, c100 = 0, c101 = 0, c110 = 0, c111 = 0, count = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        const u32 F1 = P1[j] ^ shr_var_word<NW>(P1, j, ku), F2 = P2[j] ^ shr_var_word<NW>(P2, j, ku);
        const u32 F3 = P3[j] ^ shr_var_word<NW>(P3, j, ku);
        const u32 v = V[j];
        out.V[j] = v;
        out.E[j] = ~((lo[j] ^ shr_var_word<NW>(lo, j, ku)) | (hi[j] ^ shr_var_word<NW>(hi, j, ku)));
        const u32 a1 = v & F1, a0 = v ^ a1;
        const u32 a11 = a1 & F2, a10 = a1 ^ a11, a01 = a0 & F2, a00 = a0 ^ a01;
        const u32 b111 = a11 & F3, b101 = a10 & F3, b011 = a01 & F3, b001 = a00 & F3;
        count += __popc(v);
        c111 += __popc(b111);
        c110 += __popc(a11 ^ b111);
        c101 += __popc(b101);
        c100 += __popc(a10 ^ b101);
        c011 += __popc(b011);
        c010 += __popc(a01 ^ b011);
        c001 += __popc(b001);
        c000 += __popc(a00 ^ b001);
    }
    const u32 m8 = max(max(max(c000, c001), max(c010, c011)), max(max(c100, c101), max(c110, c111)));
    // k = 64 is not bounded here (shift amounts stay below 64): never prune it
    out.ub = (k > gmax || k >= 64 || count == 0) ? ((k >= 64 && k <= gmax) ? 2.0 : 0.0) : (double) m8 / (double) count;
    {
        u32 links = 0;  // valid windows i, i+1 that share a class
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const u32 vn = j + 1 < NW ? out.V[j + 1 < NW ? j + 1 : 0] : 0u;
            links += __popc(out.V[j] & out.E[j] & alignbit(vn, out.V[j], 1u));
        }
        out.runs = count - links;
    }
    PH_ADD(PH_BOUNDS, t_ph);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <typename WT>
struct Decision {
    int kh, kl;   // target_k_high / target_k_low
    WT sh, sl;    // MAX_SEQ at those k (repeat_seq, kmer.cpp:2260-2262)
    int ek;       // k of the last class count whose canon[] / cnt[] are still in LDS (0: none)
    u32 en;       // its number of items
};

// bits j-1 for every multiple j <= 64 of k
__device__ __forceinline__ u64 multiples_mask(int k) {
    u64 m = 0;
    for (int j = k; j <= 64; j += k) m |= 1ull << (j - 1);
    return m;
}

// selection loops of k_mer_check, kmer.cpp:2221-2258, run online over ascending
// candidate k (non-candidates have frequency < LOW and can never be accepted)
// M: per-lane knowledge of lane_bounds() (lane l <-> k = MIN_MER + l); ignored when NW == 0
template <int NW, typename WT>
__device__ Decision<WT> decide(ExactSmem sm, const DevParams &P, int L, int kmin, int kmax, u64 cand, const LaneMasks<(NW > 0 ? NW : 1)> &M,
                               int lane_base = 0, int lane_span = 64) {
    constexpr bool HAVE_UB = NW > 0;
    constexpr int NWB = NW > 0 ? NW : 1;
    Decision<WT> d;
    PH_T0(t_ph);
    d.kh = d.kl = 0;
    d.sh = d.sl = 0;
    d.ek = 0;
    d.en = 0;
    double tf_low = 0.0, tf_high = 0.0;
    // closed_*: bit k-1 set <=> k is a multiple of a k already accepted in that loop.  Such a k is
    // never accepted and never moves the running frequency (kmer.cpp:2225-2236).
    u64 closed_low = 0, closed_high = 0;
    u64 todo = cand & all_k_mask(kmin, kmax);
    // Speculative skip (HAVE_UB only).  Counting the classes of a k with many runs is the expensive
    // case, and it is usually a k that cannot matter: a TTAGGG read probed at k = 5 (bound 0.5, ~45 runs)
    // just before k = 6 is accepted at 0.9.  Such a k is passed over; that is exact provided every k
    // accepted later in a selection loop the skipped k was eligible for has a frequency >= the bound of
    // the skipped k and is not a multiple of it (then the skipped k could only have raised the threshold
    // to a value the later k clears and closed multiples nobody took), and at least one such k exists
    // (else the skipped k might be the answer).  If the check fails the segment is decided again with
    // every k counted.
    constexpr u32 kHeavyRuns = 24;
    bool strict = !HAVE_UB;
    int sk_k[2] = {0, 0};
    double sk_ub[2] = {0.0, 0.0};
    u32 sk_el[2] = {0, 0}, sk_ok[2] = {0, 0};
    bool viol = false;
    const u64 todo0 = todo;
    // lane l <-> k = MIN_MER + l (as in lane_bounds); only used when HAVE_UB
    // lanes [lane_base, lane_base + lane_span) hold this segment's bounds (lane_bounds_pair: half a wave)
    const int kl = P.min_mer + (int) lane_id() - lane_base;
    const u32 klb = (u32) (kl - 1) & 63u;
    const bool my_lane = (int) lane_id() >= lane_base && (int) lane_id() < lane_base + lane_span;
again:
    for (;;) {
        const double thr_lo = P.low > tf_low ? P.low : tf_low;     // MAX(LOW_BASELINE, target_frequency_low)
        const double thr_hi = P.high > tf_high ? P.high : tf_high; // MAX(HIGH_BASELINE, target_frequency_high)
        int k;
        if (HAVE_UB) {
            // Next k the scalar loop would evaluate, found for all k at once: k still open in one of
            // the two selection loops and its bound reaches the smallest threshold it has to meet
            // (MAX <= maxbucket and IEEE division is monotone in the numerator: f <= bound < need
            // can never be accepted).  Thresholds only move when a k is accepted, after which the
            // eligibility is recomputed, so skipping is equivalent to the k-by-k walk.
            const bool lo_l = !((closed_low >> klb) & 1ull), hi_l = !((closed_high >> klb) & 1ull);
            const double need_l = lo_l ? (hi_l ? (thr_lo < thr_hi ? thr_lo : thr_hi) : thr_lo) : thr_hi;
            const u64 el = __ballot(my_lane && kl <= 64 && (lo_l || hi_l) && M.ub >= need_l) >> lane_base;
            todo &= P.min_mer > 1 ? (el << (P.min_mer - 1)) : el;
        }
        if (!todo) break;
        k = __ffsll((long long) todo);  // bit k-1 -> k, ascending
        todo &= todo - 1;
        const bool lo_open = !((closed_low >> (k - 1)) & 1ull), hi_open = !((closed_high >> (k - 1)) & 1ull);
        if (!lo_open && !hi_open) continue;
        const double need = lo_open ? (hi_open ? (thr_lo < thr_hi ? thr_lo : thr_hi) : thr_lo) : thr_hi;
        KStat<WT> st;
        // lane_bounds knows nothing about k = 64 (shift amounts stay below 64); only the 128-bit-word kernels can meet it
        constexpr bool kMay64 = sizeof(WT) > 8;
        if (HAVE_UB && !strict && (!kMay64 || k < 64)) {
            const int src = k - P.min_mer + lane_base;
            if ((u32) __builtin_amdgcn_readlane((int) M.runs, src) > kHeavyRuns && (sk_k[0] == 0 || sk_k[1] == 0)) {
                const double ub = readlane_f64(M.ub, src);
                const u32 el = ((lo_open && ub >= thr_lo) ? 1u : 0u) | ((hi_open && ub >= thr_hi) ? 2u : 0u);
                const int s = sk_k[0] == 0 ? 0 : 1;
                sk_k[s] = k;
                sk_ub[s] = ub;
                sk_el[s] = el;
                continue;
            }
        }
        if (HAVE_UB && (!kMay64 || k < 64)) {
            const int src = k - P.min_mer + lane_base;
            // the window masks of this k were computed bit-parallel by lane `src`: fetch them
            // instead of walking the windows (phase A of eval_k)
            st.count = st.maxc = st.n_items = 0;
            st.maxseq = 0;
            st.pruned = false;
            const int W = L - k + 1;
            if (W > 0) {
                __syncthreads();
                u64 *vm = sm_vmask(sm), *em = sm_emask(sm);
                // the LDS mask arrays hold cap/64 + 2 words: never write past them when the
                // instantiation's NW covers more bits than this batch's segments need
                const int nq_all = (NWB + 1) / 2, nq_fit = (int) (sm.cap / 64u) + 1;
                const int nq = nq_all < nq_fit ? nq_all : nq_fit;
                if ((int) lane_id() == src) {  // the lane that owns this k stores its masks itself: no broadcast needed
#pragma unroll
                    for (int q = 0; q < nq_all; q++) {
                        if (q >= nq) break;
                        const u32 v1 = 2 * q + 1 < NWB ? M.V[2 * q + 1 < NWB ? 2 * q + 1 : 0] : 0u;
                        const u32 e1 = 2 * q + 1 < NWB ? M.E[2 * q + 1 < NWB ? 2 * q + 1 : 0] : 0u;
                        vm[q] = ((u64) v1 << 32) | M.V[2 * q];
                        em[q] = ((u64) e1 << 32) | M.E[2 * q];
                    }
                    vm[nq] = 0;
                    em[nq] = 0;
                }
                __syncthreads();
                st = uni(eval_runs<WT>(sm, W, k));
            }
        } else {
            st = uni(eval_k<WT>(sm, L, k, need));
        }
        d.ek = st.pruned ? 0 : k;
        d.en = st.n_items;
        if (st.pruned || st.count == 0) continue;  // 0/0 = NaN fails every >=
        const double f = (double) st.maxc / (double) st.count;
        if (is_homopolymer<WT>(st.maxseq, k)) continue;
        u32 acc = 0;
        if (lo_open && f >= thr_lo) {
            d.kl = k;
            tf_low = f;
            closed_low |= multiples_mask(k);
            d.sl = st.maxseq;
            acc |= 1u;
        }
        if (hi_open && f >= thr_hi) {
            d.kh = k;
            tf_high = f;
            closed_high |= multiples_mask(k);
            d.sh = st.maxseq;
            acc |= 2u;
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (sk_k[s] && (acc & sk_el[s])) {
                if (f < sk_ub[s] || k % sk_k[s] == 0) viol = true;
                sk_ok[s] |= acc & sk_el[s];
            }
        }
    }
    if (HAVE_UB && !strict) {
#pragma unroll
        for (int s = 0; s < 2; s++)
            if (sk_k[s] && (sk_el[s] & ~sk_ok[s])) viol = true;
        if (viol) {  // decide again, counting every k
            strict = true;
            d.kh = d.kl = 0;
            d.sh = d.sl = 0;
            d.ek = 0;
            d.en = 0;
            tf_low = tf_high = 0.0;
            closed_low = closed_high = 0;
            todo = todo0;
            sk_k[0] = sk_k[1] = 0;
            goto again;
        }
    }
    PH_ADD(PH_DECIDE, t_ph);
    return d;
}

// Keep / reuse the class table a decide() left in LDS (see Decision::ek): slot = where to keep it.
// Returns k | items << 8 when kept, else 0.
template <typename WT>
__device__ __forceinline__ u32 save_classes(ExactSmem sm, const Decision<WT> &d, int slot) {
    if (!(d.ek > 0 && d.en <= kSaveItems && (d.ek == d.kh || d.ek == d.kl))) return 0u;
    if (lane_id() < d.en) {
        sm_save_canon<WT>(sm)[(u32) slot * kSaveItems + lane_id()] = sm_canon<WT>(sm)[lane_id()];
        sm_save_cnt<WT>(sm)[(u32) slot * kSaveItems + lane_id()] = sm_cnt(sm)[lane_id()];
    }
    return (u32) d.ek | (d.en << 8);
}
// bring a kept table back into canon[] / cnt[] for emit_k; returns its number of items
template <typename WT>
__device__ __forceinline__ u32 restore_classes(ExactSmem sm, u32 saved, int slot) {
    const u32 n_items = saved >> 8;
    __syncthreads();
    if (lane_id() < n_items) {
        sm_canon<WT>(sm)[lane_id()] = sm_save_canon<WT>(sm)[(u32) slot * kSaveItems + lane_id()];
        sm_cnt(sm)[lane_id()] = sm_save_cnt<WT>(sm)[(u32) slot * kSaveItems + lane_id()];
    }
    __syncthreads();
    return n_items;
}

// record the histogram of segment (already staged) at k into tables; `saved` (k | items << 8) names a
// class table of this segment kept by save_classes in `slot`, used when it is the table of this k
template <typename WT>
__device__ void record_kept(ExactSmem sm, const DevTable &T, int L, int k, u32 table_mask, bool strand_canon, u32 saved, int slot) {
    if (k <= 0 || table_mask == 0) return;
    if ((int) (saved & 255u) == k) {
        const u32 n_items = restore_classes<WT>(sm, saved, slot);
        emit_k<WT>(sm, T, n_items, k, table_mask, strand_canon);
        return;
    }
    const KStat<WT> st = uni(eval_k<WT>(sm, L, k, 0.0));
    emit_k<WT>(sm, T, st.n_items, k, table_mask, strand_canon);
}

// the same right after the decide() of this very segment: its last class count is still in canon[] / cnt[]
template <typename WT>
__device__ void record_live(ExactSmem sm, const DevTable &T, int L, int k, u32 table_mask, bool strand_canon, const Decision<WT> &d) {
    if (k <= 0 || table_mask == 0) return;
    if (d.ek == k) {
        emit_k<WT>(sm, T, d.en, k, table_mask, strand_canon);
        return;
    }
    const KStat<WT> st = uni(eval_k<WT>(sm, L, k, 0.0));
    emit_k<WT>(sm, T, st.n_items, k, table_mask, strand_canon);
}

// record the histogram of segment (already staged) at k into tables
template <typename WT>
__device__ void record(ExactSmem sm, const DevTable &T, int L, int k, u32 table_mask, bool strand_canon) {
    if (k <= 0 || table_mask == 0) return;
    PH_T0(t_ph);
    PH_CNT(PH_N_RECORD, 1);
    const KStat<WT> st = uni(eval_k<WT>(sm, L, k, 0.0));
    PH_ADD(PH_RECORD_EVAL, t_ph);
    emit_k<WT>(sm, T, st.n_items, k, table_mask, strand_canon);
}

// k_mer_target, kmer.cpp:1894-2017, on the staged whole read
template <typename WT>
__device__ void target(ExactSmem sm, const DevParams &P, const DevTable &T, int L, int k, bool want_high, bool want_low) {
    PH_T0(t_ph);
    PH_CNT(PH_N_RECORD, 1);
    const KStat<WT> st = uni(eval_k<WT>(sm, L, k, 0.0));
    PH_ADD(PH_RECORD_EVAL, t_ph);
    if (st.count == 0) return;
    const double f = is_homopolymer<WT>(st.maxseq, k) ? 0.0 : (double) st.maxc / (double) st.count;
    u32 tm = 0;
    if (want_high && f >= P.high) tm |= 1u << TREW_TABLE_BOTH_HIGH;
    if (want_low && f >= P.low) tm |= 1u << TREW_TABLE_BOTH_LOW;
    if (tm) emit_k<WT>(sm, T, st.n_items, k, tm, true);
}

// buffer_task, kmer.cpp:111-173
template <int NW, typename WT>
__device__ void run_short(ExactSmem sm, const DevParams &P, const DevTable &T, const ReadRef &rd) {
    constexpr bool UB = NW > 0;
    constexpr int NWB = NW > 0 ? NW : 1;
    const int n = (int) rd.len;
    const Segment sL = get_segment(TREW_MODE_SHORT, 0, rd.len, 0, P.min_mer, P.max_mer, P.slice_len);
    const Segment sR = get_segment(TREW_MODE_SHORT, 1, rd.len, 0, P.min_mer, P.max_mer, P.slice_len);
    const Segment sW = get_segment(TREW_MODE_SHORT, 2, rd.len, 0, P.min_mer, P.max_mer, P.slice_len);
    Decision<WT> left = {0, 0, 0, 0, 0, 0}, right = {0, 0, 0, 0, 0, 0};
    if (sL.valid || sW.valid) stage_bases(sm, rd, 0, (u32) n, 0);  // the whole read, once; segments are views
    if (sL.valid) {
        sm = view_segment(sm, 0, sL.start, rd);
        LaneMasks<NWB> mL, mR;
        const bool both = P.max_mer - P.min_mer < 32;  // the k range fits half a wave: both halves in one pass
        if (UB) {
            if (both) {
                lane_bounds_pair<NWB>(rd, sL.start, (int) sL.len, sR.start, (int) sR.len, P.min_mer, P.max_mer, mL);
            } else {
                lane_bounds<NWB>(rd, sL.start, (int) sL.len, P.min_mer, P.max_mer, mL);
                lane_bounds<NWB>(rd, sR.start, (int) sR.len, P.min_mer, P.max_mer, mR);
            }
        }
        const bool halves = UB && both;
        if (halves) mR = mL;  // the right half's bounds sit in lanes 32..63 of the same registers
        left = decide<NW, WT>(sm, P, (int) sL.len, sL.kmin, sL.kmax, ~0ull, mL, 0, halves ? 32 : 64);
        const u32 keptL = save_classes<WT>(sm, left, 0);  // a junction read records this half as it was counted here
        sm = view_segment(sm, 0, sR.start, rd);
        right = decide<NW, WT>(sm, P, (int) sR.len, sR.kmin, sR.kmax, ~0ull, mR, halves ? 32 : 0, halves ? 32 : 64);
        const u32 keptR = save_classes<WT>(sm, right, 1);
        const bool left_found = left.kh > 0 || left.kl > 0;
        const bool tgt_h = left_found && left.kh == right.kh && left.kh > 0;  // kmer.cpp:128
        const bool tgt_l = left_found && left.kl == right.kl && left.kl > 0;  // kmer.cpp:141
        // right half (staged): recorded only where the map passed was non-null
        //  - left found nothing: result.backward directly (kmer.cpp:157-158)
        //  - left found something: temp_result_right.b only when left.b == 0 (kmer.cpp:125), flushed to backward when no target
        {
            const bool rec_h = right.kh > 0 && (!left_found || (left.kh == 0));
            const bool rec_l = right.kl > 0 && (!left_found || (left.kl == 0));
            if (rec_h && rec_l && right.kh == right.kl) {
                record_kept<WT>(sm, T, (int) sR.len, right.kh, (1u << TREW_TABLE_BACKWARD_HIGH) | (1u << TREW_TABLE_BACKWARD_LOW), false, keptR, 1);
            } else {
                if (rec_h) record_kept<WT>(sm, T, (int) sR.len, right.kh, 1u << TREW_TABLE_BACKWARD_HIGH, false, keptR, 1);
                if (rec_l) record_kept<WT>(sm, T, (int) sR.len, right.kl, 1u << TREW_TABLE_BACKWARD_LOW, false, keptR, 1);
            }
        }
        if (left_found) {
            const bool rec_h = left.kh > 0 && !tgt_h;  // temp_result_left.first -> forward.first (kmer.cpp:132-134)
            const bool rec_l = left.kl > 0 && !tgt_l;
            if (rec_h || rec_l) {
                sm = view_segment(sm, 0, sL.start, rd);
                if (rec_h && rec_l && left.kh == left.kl) {
                    record_kept<WT>(sm, T, (int) sL.len, left.kh, (1u << TREW_TABLE_FORWARD_HIGH) | (1u << TREW_TABLE_FORWARD_LOW), false, keptL, 0);
                } else {
                    if (rec_h) record_kept<WT>(sm, T, (int) sL.len, left.kh, 1u << TREW_TABLE_FORWARD_HIGH, false, keptL, 0);
                    if (rec_l) record_kept<WT>(sm, T, (int) sL.len, left.kl, 1u << TREW_TABLE_FORWARD_LOW, false, keptL, 0);
                }
            }
            if (tgt_h || tgt_l) {
                sm = view_segment(sm, 0, 0, rd);
                if (tgt_h && tgt_l && left.kh == left.kl) {
                    target<WT>(sm, P, T, n, left.kh, true, true);
                } else {
                    if (tgt_h) target<WT>(sm, P, T, n, left.kh, true, false);
                    if (tgt_l) target<WT>(sm, P, T, n, left.kl, false, true);
                }
            }
        }
    }
    const bool hh = left.kh == 0 && right.kh == 0;  // kmer.cpp:165-166
    const bool lh = left.kl == 0 && right.kl == 0;
    if (sW.valid && (hh || lh)) {  // kmer.cpp:168-171
        sm = view_segment(sm, 0, 0, rd);
        LaneMasks<NWB> mW;
        if (UB) lane_bounds<NWB>(rd, 0, n, P.min_mer, P.max_mer, mW);
        const Decision<WT> w = decide<NW, WT>(sm, P, n, sW.kmin, sW.kmax, ~0ull, mW);
        const bool rec_h = hh && w.kh > 0, rec_l = lh && w.kl > 0;
        if (rec_h && rec_l && w.kh == w.kl) {
            record<WT>(sm, T, n, w.kh, (1u << TREW_TABLE_BOTH_HIGH) | (1u << TREW_TABLE_BOTH_LOW), false);
        } else {
            if (rec_h) record<WT>(sm, T, n, w.kh, 1u << TREW_TABLE_BOTH_HIGH, false);
            if (rec_l) record<WT>(sm, T, n, w.kl, 1u << TREW_TABLE_BOTH_LOW, false);
        }
    }
}

// TREW_MODE_SEGMENT: k_mer_check on the whole read, high -> table 0, low -> table 1
template <int NW, typename WT>
__device__ void run_segment(ExactSmem sm, const DevParams &P, const DevTable &T, u32 unit, const ReadRef &rd,
                            const SegResults &R) {
    constexpr bool UB = NW > 0;
    constexpr int NWB = NW > 0 ? NW : 1;
    const Segment s = get_segment(TREW_MODE_SEGMENT, 0, rd.len, 0, P.min_mer, P.max_mer, P.slice_len);
    if (!s.valid) return;
    sm = load_segment(sm, rd, 0, s.len);
    LaneMasks<NWB> m;
    if (UB) lane_bounds<NWB>(rd, 0, (int) s.len, P.min_mer, P.max_mer, m);
    const Decision<WT> d = decide<NW, WT>(sm, P, (int) s.len, s.kmin, s.kmax, ~0ull, m);
    if (d.kh > 0 && d.kh == d.kl) {
        record<WT>(sm, T, (int) s.len, d.kh, (1u << TREW_TABLE_FORWARD_HIGH) | (1u << TREW_TABLE_FORWARD_LOW), false);
    } else {
        record<WT>(sm, T, (int) s.len, d.kh, 1u << TREW_TABLE_FORWARD_HIGH, false);
        record<WT>(sm, T, (int) s.len, d.kl, 1u << TREW_TABLE_FORWARD_LOW, false);
    }
    if (lane_id() == 0 && R.k_high) {
        R.k_high[unit] = d.kh;
        R.k_low[unit] = d.kl;
        R.seq_high[unit] = (u64) d.sh;
        R.seq_low[unit] = (u64) d.sl;
        if (R.seq_high_hi) {
            R.seq_high_hi[unit] = sizeof(WT) > 8 ? (u64) (d.sh >> (8 * sizeof(WT) - 64)) : 0ull;
            R.seq_low_hi[unit] = sizeof(WT) > 8 ? (u64) (d.sl >> (8 * sizeof(WT) - 64)) : 0ull;
        }
    }
}

// ------------------------------------------------------------------ long reads
// slice t (1-based) of buffer_task_long, kmer.cpp:790-798: SLICE_LENGTH bases each, the
// remainder len % SLICE_LENGTH goes to slice mid = (snum+1)/2
__device__ __forceinline__ void long_slice(int t, int mid, int bonus, int SL, u32 &start, u32 &len) {
    start = (u32) ((t - 1) * SL + (t > mid ? bonus : 0));
    len = (u32) (SL + (t == mid ? bonus : 0));
}

// buffer_task_long, kmer.cpp:785-871.  The forward walk accumulates into temp_result_left,
// whose destination (both, strand-canonical / forward) is only known when the walk ends, so
// it runs twice: pass 1 decides, pass 2 re-decides the recorded slices and emits.  The
// backward walk records straight into result.backward (kmer.cpp:840).
template <int NW, typename WT>
__device__ void run_long(ExactSmem sm, const DevParams &P, const DevBatch &B, const DevTable &T, u32 unit) {
    constexpr bool UB = NW > 0;
    constexpr int NWB = NW > 0 ? NW : 1;
    const ReadRef rd = uni(get_read(B, rfl(unit)));
    const int SL = P.slice_len;
    const int len = (int) rd.len;
    const int snum = len / SL;
    if (snum <= 0) return;  // read_fastq_long_thread drops reads shorter than SLICE_LENGTH (kmer.cpp:1184)
    const int mid = (snum + 1) / 2, bonus = len % SL;
    const u64 allk = all_k_mask(P.min_mer, P.max_mer);
    // bounds of two neighbouring slices share one lane pass (lanes 0..31 / 32..63) when the k range fits
    // half a wave: a walk that continues finds the next slice's bounds already there
    constexpr int NS = NW > 5 ? 5 : NWB;
    LaneMasks<NS> mp;
    int mp_lo = 0, mp_hi = 0;  // slices whose bounds sit in the low / high half of mp (0: none)
    auto slice_decide = [&](int t, u64 cand, int dir) {
        u32 st, sl;
        long_slice(t, mid, bonus, SL, st, sl);
        sm = load_segment(sm, rd, st, sl);
        if (NW > 5 && sl <= 159u) {  // every slice but the middle one is SLICE_LENGTH long: half the mask words
            if (P.max_mer - P.min_mer < 32) {
                if (t != mp_lo && t != mp_hi) {
                    const int tn = t + dir;
                    u32 stn = st, sln = sl;
                    int other = 0;
                    if (tn >= 1 && tn <= snum) {
                        long_slice(tn, mid, bonus, SL, stn, sln);
                        if (sln <= 159u) other = tn;
                    }
                    if (!other) {
                        stn = st;
                        sln = sl;
                    }
                    lane_bounds_pair<NS>(rd, st, (int) sl, stn, (int) sln, P.min_mer, P.max_mer, mp);
                    mp_lo = t;
                    mp_hi = other;
                }
                return decide<(NW > 5 ? 5 : NW), WT>(sm, P, (int) sl, P.min_mer, P.max_mer, cand, mp, t == mp_lo ? 0 : 32, 32);
            }
            LaneMasks<NS> m5;
            lane_bounds<NS>(rd, st, (int) sl, P.min_mer, P.max_mer, m5);
            return decide<(NW > 5 ? 5 : NW), WT>(sm, P, (int) sl, P.min_mer, P.max_mer, cand, m5);
        }
        LaneMasks<NWB> m;
        if (UB) lane_bounds<NWB>(rd, st, (int) sl, P.min_mer, P.max_mer, m);
        return decide<NW, WT>(sm, P, (int) sl, P.min_mer, P.max_mer, cand, m);
    };
    auto slice_len = [&](int t) { return (int) (SL + (t == mid ? bonus : 0)); };
    // pass 1: forward chain (kmer.cpp:797-817)
    int si[2] = {1, 1}, kmer[2] = {0, 0}, last_rec[2] = {0, 0};
    bool rend[2] = {false, false};
    // decisions of the first 64 slices, one per lane (kh | kl << 8), so that pass 2 does not decide them again
    u32 dcache = 0;
    for (int ti = 1; ti <= snum && (!rend[0] || !rend[1]); ti++) {
        const Decision<WT> d = slice_decide(ti, ti == 1 ? ~0ull : (ti == snum ? ~0ull : allk), 1);
        if ((int) lane_id() == ti - 1) dcache = (u32) d.kh | ((u32) d.kl << 8);
        const int tk[2] = {d.kh, d.kl};
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if (!rend[b]) last_rec[b] = ti;  // this slice was recorded into temp_result_left[b]
            if (!rend[b] && tk[b] > 0 && (kmer[b] == tk[b] || ti == 1)) {
                si[b] += 1;
                kmer[b] = tk[b];
            } else {
                rend[b] = true;
            }
        }
    }
    // pass 2: emit temp_result_left: strand-canonical into both when every slice chained
    // (kmer.cpp:819-830), else as is into forward (kmer.cpp:858-867)
    {
        const int upto = last_rec[0] > last_rec[1] ? last_rec[0] : last_rec[1];
        const bool canon_h = si[0] == snum + 1, canon_l = si[1] == snum + 1;
        const u32 th = canon_h ? TREW_TABLE_BOTH_HIGH : TREW_TABLE_FORWARD_HIGH;
        const u32 tl = canon_l ? TREW_TABLE_BOTH_LOW : TREW_TABLE_FORWARD_LOW;
        for (int ti = 1; ti <= upto; ti++) {
            Decision<WT> d;
            if (ti <= 64) {
                const u32 c = (u32) __builtin_amdgcn_readlane((int) dcache, ti - 1);
                d.kh = (int) (c & 255u);
                d.kl = (int) (c >> 8);
                d.sh = d.sl = 0;
                d.ek = 0;
                d.en = 0;
                if ((ti <= last_rec[0] && d.kh > 0) || (ti <= last_rec[1] && d.kl > 0)) {  // record() wants the slice staged
                    u32 st, sl;
                    long_slice(ti, mid, bonus, SL, st, sl);
                    sm = load_segment(sm, rd, st, sl);
                }
            } else {
                d = slice_decide(ti, ti == 1 ? ~0ull : (ti == snum ? ~0ull : allk), 1);
            }
            const bool rh = ti <= last_rec[0] && d.kh > 0, rl = ti <= last_rec[1] && d.kl > 0;
            if (rh && rl && d.kh == d.kl && canon_h == canon_l) {
                record<WT>(sm, T, slice_len(ti), d.kh, (1u << th) | (1u << tl), canon_h);
            } else {
                if (rh) record<WT>(sm, T, slice_len(ti), d.kh, 1u << th, canon_h);
                if (rl) record<WT>(sm, T, slice_len(ti), d.kl, 1u << tl, canon_l);
            }
        }
    }
    // backward chain (kmer.cpp:832-856)
    if (si[0] <= snum || si[1] <= snum) {
        int sj[2] = {snum, snum};
        kmer[0] = kmer[1] = 0;
        rend[0] = rend[1] = false;
        for (int tj = snum; (!rend[0] || !rend[1]) && tj >= 1; tj--) {
            const Decision<WT> d = slice_decide(tj, tj == snum ? ~0ull : (tj == 1 ? ~0ull : allk), -1);
            const bool rh = !rend[0] && d.kh > 0, rl = !rend[1] && d.kl > 0;
            if (rh && rl && d.kh == d.kl) {
                record_live<WT>(sm, T, slice_len(tj), d.kh, (1u << TREW_TABLE_BACKWARD_HIGH) | (1u << TREW_TABLE_BACKWARD_LOW), false, d);
            } else {  // two different k: the first record may count again and overwrite the live table
                Decision<WT> d2 = d;
                if (rh) {
                    record_live<WT>(sm, T, slice_len(tj), d.kh, 1u << TREW_TABLE_BACKWARD_HIGH, false, d2);
                    if (d2.ek != d.kh) d2.ek = 0;
                }
                if (rl) record_live<WT>(sm, T, slice_len(tj), d.kl, 1u << TREW_TABLE_BACKWARD_LOW, false, d2);
            }
            const int tk[2] = {d.kh, d.kl};
#pragma unroll
            for (int b = 0; b < 2; b++) {
                if (sj[b] >= si[b] && !rend[b] && tk[b] > 0 && (kmer[b] == tk[b] || tj == snum)) {
                    sj[b] -= 1;
                    kmer[b] = tk[b];
                } else {
                    rend[b] = true;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ paired reads
// deferred emission: (slot, k, baseline) recorded into temp_result_left (temp 0) or _right (temp 1)
__device__ __forceinline__ u32 pack_intent(int slot, int k, int b, int temp) {
    return (u32) slot | ((u32) k << 3) | ((u32) b << 10) | ((u32) temp << 11) | (1u << 12);
}

// destination byte of flush(): temp (0 left, 1 right), baseline (0 high, 1 low) -> table
__device__ __forceinline__ u32 dest(int temp, int b, int table) { return (1u << table) << (8 * (2 * temp + b)); }

// get_dir_seq, kmer.cpp:307-313
template <typename WT>
__device__ __forceinline__ WT dir_seq(int i, int k, WT seq, bool is_for) {
    if ((i <= 2) == is_for) return seq;
    return min_rotation<WT>(revcomp(seq, k), k);
}

// buffer_task_pair, kmer.cpp:322-507, with the 128-bit twin's clear of temp_result_left after
// the whole-read block (kmer.cpp:722-723; SURVEY G1 -- the one documented divergence from the
// 64-bit branch, whose stale map makes results depend on thread scheduling).
template <int NW, typename WT>
__device__ void run_pair(ExactSmem sm, const DevParams &P, const DevBatch &B, const DevTable &T, u32 unit_in) {
    constexpr bool UB = NW > 0;
    constexpr int NWB = NW > 0 ? NW : 1;
    const u64 unit = rfl(unit_in);
    PH_T0(t_st);
    const ReadRef r0 = uni(stage_read(sm, uni(get_read(B, 2ull * unit)), 0));
    const ReadRef r1 = uni(stage_read(sm, uni(get_read(B, 2ull * unit + 1)), 1));
    PH_ADD(PH_PAIR_STAGE, t_st);
    PH_CNT(PH_N_READS, 1);
    const int n1 = (int) r0.len, n2 = (int) r1.len;
    const int n = n1 < n2 ? n1 : n2;
    if (2 * P.min_mer > n) return;
    stage_bases(sm, r0, 0, (u32) n1, 0);  // both mates, once; every segment is a view
    stage_bases(sm, r1, 0, (u32) n2, 1);
    // the intent list lives in a register, entry i in lane i: reading it back is a readlane, not an
    // LDS round trip (the list is scanned quadratically by flush(); in LDS that was 20 % of the kernel)
    u32 my_intent = 0;
    u32 n_int = 0;
    const u32 lane = lane_id();
    auto seg_of = [&](int slot) { return get_segment(TREW_MODE_PAIR, slot, (u32) n1, (u32) n2, P.min_mer, P.max_mer, P.slice_len); };
    // k_mer_check is a pure function of the segment: the backward chain reuses what the forward
    // chain decided (slot s cached in lane s)
    u32 dc_k = 0, dc_have = 0, dc_saved = 0;  // dc_saved: k | items << 8 of the class table kept for the slot
    WT dc_sh = 0, dc_sl = 0;
    LaneMasks<NWB> mp;  // bounds of both halves of mate mp_pair
    int mp_pair = -1;
    auto seg_decide = [&](int slot) {
        if ((dc_have >> slot) & 1u) {
            Decision<WT> d;
            const u32 c = (u32) __builtin_amdgcn_readlane((int) dc_k, slot);
            d.kh = (int) (c & 255u);
            d.kl = (int) (c >> 8);
            d.sh = readlane_word(dc_sh, slot);
            d.sl = readlane_word(dc_sl, slot);
            d.ek = 0;
            d.en = 0;
            return d;
        }
        const Segment sg = seg_of(slot);
        const ReadRef &r = sg.mate ? r1 : r0;
        const ExactSmem sv = view_segment(sm, sg.mate, sg.start, r);
        Decision<WT> d;
        if (UB && slot < 4 && P.max_mer - P.min_mer < 32) {
            // the two halves of one mate share a lane pass (lanes 0..31 / 32..63); the chains visit them back to back
            const int pr = slot >> 1;
            if (mp_pair != pr) {
                const Segment a = seg_of(2 * pr), b = seg_of(2 * pr + 1);
                lane_bounds_pair<NWB>(r, a.start, (int) a.len, b.start, (int) b.len, P.min_mer, P.max_mer, mp);
                mp_pair = pr;
            }
            d = decide<NW, WT>(sv, P, (int) sg.len, sg.kmin, sg.kmax, ~0ull, mp, (slot & 1) ? 32 : 0, 32);
        } else {
            LaneMasks<NWB> m;
            if (UB) lane_bounds<NWB>(r, sg.start, (int) sg.len, P.min_mer, P.max_mer, m);
            d = decide<NW, WT>(sv, P, (int) sg.len, sg.kmin, sg.kmax, ~0ull, m);
        }
        // The classes of the last evaluated k are still in LDS.  When that k is the accepted one (the usual
        // case: nothing after it passes its bound) keep them, so that flush() adds them without counting again.
        const u32 saved = save_classes<WT>(sm, d, slot);
        if ((int) lane == slot) {
            dc_k = (u32) d.kh | ((u32) d.kl << 8);
            dc_sh = d.sh;
            dc_sl = d.sl;
            dc_saved = saved;
        }
        dc_have |= 1u << slot;
        return d;
    };
    auto add_intent = [&](int slot, int k, int b, int temp) {
        if (k > 0) {
            if (n_int < 32) {
                if (lane == n_int) my_intent = pack_intent(slot, k, b, temp);
                n_int++;
            } else if (lane == 0) {
                atomicAdd(&T.overflow[kDiagIntentDrop], 1u);  // at most 20 per pair today; a change that breaks this fails loudly in collect
            }
        }
    };
    // resolve the intent list: destination tables per (temp, baseline); plain[] rotation-canonical
    // keys, canon[] strand-canonical keys.  Only entries whose baseline is in bmask are consumed;
    // intents with equal (slot, k) share one evaluation.
    // plain/canon: byte (2*temp + baseline) = table mask (packed so nothing is indexed in scratch)
    auto flush = [&](u32 plain, u32 canon, u32 bmask, bool clear) {
        PH_T0(t_fl);
        for (u32 i = 0; i < n_int; i++) {
            const u32 e = (u32) __builtin_amdgcn_readlane((int) my_intent, (int) i);
            if (!(e >> 12) || !((bmask >> ((e >> 10) & 1u)) & 1u)) continue;  // consumed earlier / other baseline
            const int slot = (int) (e & 7u), k = (int) ((e >> 3) & 127u);
            u32 mp = 0, mc = 0;
            for (u32 j = i; j < n_int; j++) {
                const u32 f = (u32) __builtin_amdgcn_readlane((int) my_intent, (int) j);
                const u32 fb = (f >> 10) & 1u;
                if ((f >> 12) && ((bmask >> fb) & 1u) && (int) (f & 7u) == slot && (int) ((f >> 3) & 127u) == k) {
                    const u32 temp = (f >> 11) & 1u;
                    mp |= (plain >> (8u * (2u * temp + fb))) & 0xffu;
                    mc |= (canon >> (8u * (2u * temp + fb))) & 0xffu;
                    if (lane == j) my_intent = 0;
                }
            }
            if (mp | mc) {
                const Segment sg = seg_of(slot);
                const ExactSmem sv = view_segment(sm, sg.mate, sg.start, sg.mate ? r1 : r0);
                const u32 sk = (u32) __builtin_amdgcn_readlane((int) dc_saved, slot);
                u32 n_items;
                if ((int) (sk & 255u) == k) {  // counted when the slot was decided
                    n_items = restore_classes<WT>(sm, sk, slot);
                } else {
                    n_items = uni(eval_k<WT>(sv, (int) sg.len, k, 0.0)).n_items;
                }
                if (mp) emit_k<WT>(sv, T, n_items, k, mp, false);
                if (mc) emit_k<WT>(sv, T, n_items, k, mc, true);
            }
        }
        if (clear) {
            n_int = 0;
            my_intent = 0;
        }
        PH_ADD(PH_PAIR_FLUSH, t_fl);
    };
    int lef_k[2] = {0, 0}, kmer[2] = {0, 0};
    WT kseq[2] = {0, 0};
    if (4 * P.min_mer <= n) {
        // fragment order R1-left, R1-right, R2-right, R2-left = slots 0..3 (kmer.cpp:338-340)
        const int snum = 4;
        int si[2] = {1, 1};
        bool rend[2] = {false, false};
        PH_T0(t_fw);
        for (int ti = 1; ti <= snum && (!rend[0] || !rend[1]); ti++) {  // kmer.cpp:347-374
            const Decision<WT> d = seg_decide(ti - 1);
            const int tk[2] = {d.kh, d.kl};
            const WT ts[2] = {d.sh, d.sl};
#pragma unroll
            for (int b = 0; b < 2; b++) {
                if (!rend[b]) add_intent(ti - 1, tk[b], b, ti <= 2 ? 0 : 1);
                if (!rend[b] && tk[b] > 0 && ((kmer[b] == tk[b] && kseq[b] == dir_seq<WT>(ti, tk[b], ts[b], true)) || ti == 1)) {
                    si[b] += 1;
                    kmer[b] = tk[b];
                    if (ti == 1) kseq[b] = ts[b];
                } else {
                    rend[b] = true;
                }
            }
        }
        PH_ADD(PH_PAIR_FWD, t_fw);
        lef_k[0] = kmer[0];
        lef_k[1] = kmer[1];
        // all four segments chained: both temps -> both, strand-canonical (kmer.cpp:378-399).
        // The reference adds them BEFORE the backward chain refills the temps, so resolve that
        // baseline now; the other baseline's entries stay pending.
        {
            u32 canon = 0, bmask = 0;
            if (si[0] == snum + 1) {
                canon |= dest(0, 0, TREW_TABLE_BOTH_HIGH) | dest(1, 0, TREW_TABLE_BOTH_HIGH);
                bmask |= 1u;
            }
            if (si[1] == snum + 1) {
                canon |= dest(0, 1, TREW_TABLE_BOTH_LOW) | dest(1, 1, TREW_TABLE_BOTH_LOW);
                bmask |= 2u;
            }
            if (bmask) flush(0u, canon, bmask, false);
        }
        if (si[0] <= snum || si[1] <= snum) {  // backward chain, kmer.cpp:401-436
            int sj[2] = {snum, snum};
            kmer[0] = kmer[1] = 0;
            rend[0] = rend[1] = false;
            PH_T0(t_bw);
            for (int tj = snum; (!rend[0] || !rend[1]) && tj >= 1; tj--) {
                PH_CNT(PH_N_RECORD, 1);
                const Decision<WT> d = seg_decide(tj - 1);
                const int tk[2] = {d.kh, d.kl};
                const WT ts[2] = {d.sh, d.sl};
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    if (!rend[b]) add_intent(tj - 1, tk[b], b, tj <= 2 ? 1 : 0);
                    if (sj[b] >= si[b] && !rend[b] && tk[b] > 0 &&
                        ((kmer[b] == tk[b] && kseq[b] == dir_seq<WT>(tj, tk[b], ts[b], false)) || tj == snum)) {
                        sj[b] -= 1;
                        kmer[b] = tk[b];
                        if (tj == snum) kseq[b] = ts[b];
                    } else {
                        rend[b] = true;
                    }
                }
            }
            PH_ADD(PH_PAIR_BWD, t_bw);
        }
        {  // kmer.cpp:438-455: temp_left -> forward, temp_right -> backward for a baseline that did not complete
            u32 plain = 0;
            if (si[0] <= snum) plain |= dest(0, 0, TREW_TABLE_FORWARD_HIGH) | dest(1, 0, TREW_TABLE_BACKWARD_HIGH);
            if (si[1] <= snum) plain |= dest(0, 1, TREW_TABLE_FORWARD_LOW) | dest(1, 1, TREW_TABLE_BACKWARD_LOW);
            flush(plain, 0u, 3u, true);
        }
    }
    if (4 * P.max_mer > n) {  // whole-read block, kmer.cpp:467-505
        PH_T0(t_wh);
        Decision<WT> lt = {0, 0, 0, 0, 0, 0}, rt = {0, 0, 0, 0, 0, 0};
        if (lef_k[0] == 0 || lef_k[1] == 0) {
            lt = seg_decide(4);
            if (lef_k[0] == 0) add_intent(4, lt.kh, 0, 0);
            if (lef_k[1] == 0) add_intent(4, lt.kl, 1, 0);
        }
        if (kmer[0] == 0 || kmer[1] == 0) {
            rt = seg_decide(5);
            if (kmer[0] == 0) add_intent(5, rt.kh, 0, 0);
            if (kmer[1] == 0) add_intent(5, rt.kl, 1, 0);
        }
        const u32 plain = dest(0, 0, TREW_TABLE_FORWARD_HIGH) | dest(0, 1, TREW_TABLE_FORWARD_LOW);
        u32 canon = 0;
        if (lef_k[0] == 0 && kmer[0] == 0 && lt.kh == rt.kh && lt.kh > 0 && lt.sh == min_rotation<WT>(revcomp(rt.sh, rt.kh), rt.kh))
            canon |= dest(0, 0, TREW_TABLE_BOTH_HIGH);
        if (lef_k[1] == 0 && kmer[1] == 0 && lt.kl == rt.kl && lt.kl > 0 && lt.sl == min_rotation<WT>(revcomp(rt.sl, rt.kl), rt.kl))
            canon |= dest(0, 1, TREW_TABLE_BOTH_LOW);
        flush(plain, canon, 3u, true);
        PH_ADD(PH_PAIR_WHOLE, t_wh);
    }
}


// NW > 0: every staged segment fits 32*NW-1 bases and decide() prunes with lane_bounds<NW>;
// NW == 0: long segments, pruning happens inside eval_k instead.
template <int NW, int MODE, typename WT>
__global__ __launch_bounds__(64, (NW >= 10 ? 4 : 6)) void exact_kernel(DevParams P, DevBatch B, DevTable T, const u32 *wl,
                                                   u32 *wl_count, u32 *wl_count_next, u32 wl_cap, SegResults R, u32 cap, u32 rawwords) {
    // A slot owns two counter blocks and alternates between them: while this launch consumes one, its first wave
    // clears the other for the slot's next submit (stream order puts that submit's prefilter after this kernel), so
    // a submit needs no memset call.
    if (blockIdx.x == 0 && wl_count_next) {
        for (u32 i = lane_id(); i < (1u + 8u) * 32u; i += 64u) wl_count_next[i] = 0u;
    }
    ExactSmem sm;
    sm.cap = cap;
    sm.rawwords = rawwords;
    sm.s0 = 0;
    sm.rs = sm.rnw = 0;
    sm.rw = nullptr;
    u32 n = wl_count[0];
    n = n < wl_cap ? n : wl_cap;
#ifdef TREW_PHASE_PROFILE
    if (lane_id() < 32) ph_lds()[lane_id()] = 0;
    __syncthreads();
#endif
    PH_T0(t_total);
    if (P.flags & TREW_FLAG_DEBUG_POISON_LDS) {  // tests: nothing may depend on what a previous kernel left in LDS
        const u32 nb = exact_lds_bytes(cap, rawwords, sizeof(WT));
        for (u32 i = lane_id() * 4u; i < nb; i += 256u) *(u32 *) (lds0() + i) = 0xA5C3F00Du ^ (i * 2654435761u);
        __syncthreads();
    }
    cache_clear(sm);
    // dynamic self-scheduling: reads differ 10x in cost, so waves pull work from device counters
    // instead of a static stride.  One returning atomic on a single word saturates at ~88
    // dequeues/us on MI355X (MI355X_MICROARCH.md, row "dequeue") -- 2 ms for 176 k reads -- so the
    // queue head is sharded 8 ways (own cache line each; chunk c of shard s = global chunk 8c+s)
    // and a wave whose shard runs dry steals from the next one.
    // A chunk is kChunk consecutive worklist items: their reads are fetched together so that the
    // dependent global round trips (queue head -> worklist entry -> read words) are paid once
    // per chunk, not once per read (measured: 0.40 ms of a 1.3 ms launch was this latency chain).
    constexpr u32 kChunk = 4, kShards = 8, kHeadStride = 32;
    u32 *heads = wl_count + kHeadStride;
    const u32 my = blockIdx.x & (kShards - 1);
    const u32 lane = lane_id();
    for (u32 attempt = 0; attempt < kShards; attempt++) {
        const u32 sh = (my + attempt) & (kShards - 1);
        for (;;) {
            PH_T0(t_stage);
            u32 c = 0;
            if (lane == 0) c = atomicAdd(&heads[sh * kHeadStride], 1u);
            c = rfl(c);
            const u64 w0l = ((u64) c * kShards + sh) * kChunk;
            if (w0l >= n) break;
            const u32 w0 = (u32) w0l;
            const u32 nit = (w0 + kChunk < n ? w0 + kChunk : n) - w0;
            u32 units[kChunk];
#pragma unroll
            for (u32 t = 0; t < kChunk; t++) units[t] = t < nit ? wl[w0 + t] : 0u;
            // one instantiation per mode: the short-read kernel does not carry the pair driver's registers
            if (MODE == TREW_MODE_SHORT || MODE == TREW_MODE_SEGMENT) {
                ReadRef rds[kChunk];
                u32 head[kChunk];  // word `lane` of each read
#pragma unroll
                for (u32 t = 0; t < kChunk; t++) {
                    rds[t].w = B.words;
                    rds[t].len = 0;
                    rds[t].nw = 0;
                    if (t < nit) rds[t] = get_read(B, units[t]);
                }
#pragma unroll
                for (u32 t = 0; t < kChunk; t++) head[t] = lane < 3u * rds[t].nw ? rds[t].w[lane] : 0u;
                __syncthreads();  // the previous chunk no longer reads raw[]
                u32 *meta = sm_intent(sm);  // per item: length, unit, word offset of the read (lo, hi), staged flag
#pragma unroll
                for (u32 t = 0; t < kChunk; t++) {
                    const u32 nwords = 3u * rds[t].nw;
                    const bool fits = nwords <= sm.rawwords;
                    if (fits) {
                        u32 *dst = sm_raw(sm) + t * sm.rawwords;
                        if (lane < nwords) dst[lane] = head[t];
                        for (u32 j = lane + 64u; j < nwords; j += 64u) dst[j] = rds[t].w[j];
                    }
                    if (lane == 0) {
                        const u64 off = (u64) (rds[t].w - B.words);
                        meta[5 * t + 0] = rds[t].len;
                        meta[5 * t + 1] = units[t];
                        meta[5 * t + 2] = (u32) off;
                        meta[5 * t + 3] = (u32) (off >> 32);
                        meta[5 * t + 4] = fits ? 1u : 0u;
                    }
                }
                __syncthreads();
                PH_ADD(PH_STAGE, t_stage);
                PH_CNT(PH_N_READS, nit);
                for (u32 t = 0; t < nit; t++) {  // not unrolled: one copy of the driver
                    ReadRef rd;
                    rd.len = rfl(meta[5 * t + 0]);
                    rd.nw = (rd.len + 31u) >> 5;
                    const u32 unit = rfl(meta[5 * t + 1]);
                    const u64 off = ((u64) rfl(meta[5 * t + 3]) << 32) | rfl(meta[5 * t + 2]);
                    rd.w = rfl(meta[5 * t + 4]) ? (const u32 *) (sm_raw(sm) + t * sm.rawwords) : B.words + off;
                    if (MODE == TREW_MODE_SHORT)
                        run_short<NW, WT>(sm, P, T, rd);
                    else
                        run_segment<NW, WT>(sm, P, T, unit, rd, R);
                    __syncthreads();
                }
            } else {
#pragma unroll
                for (u32 t = 0; t < kChunk; t++) {
                    if (t < nit) {
                        if (MODE == TREW_MODE_LONG)
                            run_long<NW, WT>(sm, P, B, T, units[t]);
                        else
                            run_pair<NW, WT>(sm, P, B, T, units[t]);
                        __syncthreads();
                    }
                }
            }
        }
    }
    {
        PH_T0(t_flush);
        cache_flush(sm, T);
        PH_ADD(PH_FLUSH, t_flush);
    }
    PH_ADD(PH_TOTAL, t_total);
#ifdef TREW_PHASE_PROFILE
    __syncthreads();
    if (lane_id() < 32) atomicAdd(&g_phase[lane_id()], ph_lds()[lane_id()]);
#endif
}

#ifdef TREW_PHASE_PROFILE
extern "C" int trew_debug_phases(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// ------------------------------------------------------------------ table maintenance
__global__ void table_add_rows_kernel(DevTable T, const trew_hip_row *rows, u64 n) {
    const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const trew_hip_row r = rows[i];
    if (r.k < 1 || r.k > 64 || r.table < 0 || r.table >= TREW_NUM_TABLES || (r.k <= 32 && r.word_hi)) {
        atomicAdd(&T.overflow[kDiagBadRow], 1u);
        return;
    }
    if (r.count) table_add(T, r.table, r.k, ((u128) r.word_hi << 64) | r.word_lo, r.count);
}

// compaction of the sparse table into rows (collect): one atomic per occupied slot
__global__ void table_compact_kernel(DevTable T, u64 n_slots, int table, trew_hip_row *rows, u64 cap, unsigned long long *n_rows) {
    const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) {  // wide slots follow the narrow ones
        const u64 j = i - n_slots;
        const DevWide Wd = *T.wide;
        if (j >= (1ull << Wd.wide_log2_slots)) return;
        const u64 tag = Wd.wtag[j];
        if (!(tag >> 63)) return;
        const int t = (int) ((tag >> 59) & 7ull);
        if (table >= 0 && t != table) return;
        const u64 at = atomicAdd(n_rows, 1ull);
        if (at < cap) {
            trew_hip_row r;
            r.k = (int32_t) ((tag >> 52) & 127ull);
            r.table = t;
            r.word_lo = Wd.wlo[j];
            r.word_hi = Wd.whi[j];
            r.count = Wd.wcount[j];
            rows[at] = r;
        }
        return;
    }
    const u64 key = T.keys[i];
    if (!key) return;
    const int t = (int) ((key >> 60) & 7ull);
    if (table >= 0 && t != table) return;
    const u64 at = atomicAdd(n_rows, 1ull);
    if (at < cap) {
        trew_hip_row r;
        r.k = (int32_t) ((key >> 55) & 31ull) + 1;
        r.table = t;
        r.word_lo = ((key & ((1ull << 55) - 1ull)) << kTablePartBits) | (i >> T.log2_part_slots);
        r.word_hi = 0;
        r.count = T.counts[i];
        rows[at] = r;
    }
}

// ------------------------------------------------------------------ synthetic generators
__global__ void synth_short_kernel(u64 seed, u64 first_read, u64 n_reads, u32 read_len, u32 *words) {
    const u32 nw = (read_len + 31u) >> 5;
    const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reads * nw) return;
    const u64 r = t / nw;
    const u32 j = (u32) (t % nw);
    const trew_synth::ReadClass c = trew_synth::read_class(seed, first_read + r);
    u32 lo = 0, hi = 0, nm = 0;
    for (u32 i = 0; i < 32; i++) {
        const u32 pos = 32u * j + i;
        if (pos >= read_len) break;
        const int b = trew_synth::short_base(seed, first_read + r, c, pos, read_len);
        if (b > 3) {
            nm |= 1u << i;
        } else {
            lo |= (u32) (b & 1) << i;
            hi |= (u32) (b >> 1) << i;
        }
    }
    u32 *o = words + (r * nw + j) * 3ull;
    o[0] = lo;
    o[1] = hi;
    o[2] = nm;
}

__global__ void synth_pair_kernel(u64 seed, u64 first_pair, u64 n_pairs, u32 read_len, u32 *words) {
    const u32 nw = (read_len + 31u) >> 5;
    const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs * 2ull * nw) return;
    const u64 rr = t / nw;  // read index: 2*pair + mate
    const u32 j = (u32) (t % nw);
    const u64 pair = rr >> 1;
    const int mate = (int) (rr & 1ull);
    const trew_synth::ReadClass c = trew_synth::read_class(seed, first_pair + pair);
    u32 lo = 0, hi = 0, nm = 0;
    for (u32 i = 0; i < 32; i++) {
        const u32 pos = 32u * j + i;
        if (pos >= read_len) break;
        const int b = trew_synth::pair_base(seed, first_pair + pair, c, mate, pos, read_len);
        if (b > 3) {
            nm |= 1u << i;
        } else {
            lo |= (u32) (b & 1) << i;
            hi |= (u32) (b >> 1) << i;
        }
    }
    u32 *o = words + (rr * nw + j) * 3ull;
    o[0] = lo;
    o[1] = hi;
    o[2] = nm;
}

// one block per long read; lengths[] / offsets[] (u32 word offsets) are precomputed on the host
__global__ void synth_long_kernel(u64 seed, u64 first_read, u64 n_reads, const u32 *qtable, const u32 *offsets, u32 *words) {
    const u64 r = blockIdx.x;
    if (r >= n_reads) return;
    const trew_synth::LongClass c = trew_synth::long_class(seed, first_read + r, qtable);
    const u32 nw = (c.len + 31u) >> 5;
    u32 *o = words + offsets[r];
    for (u32 j = threadIdx.x; j < nw; j += blockDim.x) {
        u32 lo = 0, hi = 0;
        for (u32 i = 0; i < 32; i++) {
            const u32 pos = 32u * j + i;
            if (pos >= c.len) break;
            const int b = trew_synth::long_base(seed, first_read + r, c, pos);
            lo |= (u32) (b & 1) << i;
            hi |= (u32) (b >> 1) << i;
        }
        o[3 * j + 0] = lo;
        o[3 * j + 1] = hi;
        o[3 * j + 2] = 0;
    }
}

// ------------------------------------------------------------------ launchers
int pick_nw(u32 max_seg_len) {
    if (max_seg_len <= 95) return 3;
    if (max_seg_len <= 159) return 5;
    if (max_seg_len <= 319) return 10;
    return 32;
}

// Pass thresholds of the uniform-geometry fast path for one batch geometry: table[slot * kThrRow + k - 1] = {ithr, jthr}.
// Host side (single-precision multiply, the same IEEE operation the general path performs on the device).
void fill_thresholds(const DevParams &P, u32 uniform_length, int2 *table) {
    const int nslots = mode_slots(P.mode);
    for (int slot = 0; slot < kMaxSlots; slot++) {
        const Segment sg = get_segment(P.mode, slot, uniform_length, uniform_length, P.min_mer, P.max_mer, P.slice_len);
        for (int k = 1; k <= kThrRow; k++) {
            const int W = (int) sg.len - k + 1;
            int2 th;
            th.x = 0x7fffffff;  // nothing passes
            th.y = -1;
            if (slot < nslots && sg.valid && W > 0) {
                const volatile float prod = (float) W * P.lowf;  // volatile: no contraction, no extended precision
                th.x = (int) floorf(prod) + 1;
                th.y = W - th.x;
            }
            table[slot * kThrRow + k - 1] = th;
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
    // Persistent blocks with a static, grid-strided share of the reads each; twice the resident number of blocks
    // (78 VGPRs -> 6 waves per SIMD for 150-bp reads), so that CUs whose blocks finish early pick up another one.
    static thread_local kern_t cached_fn = nullptr;
    static thread_local int cached_per_cu = 0;
    int per_cu = cached_per_cu;
    if (cached_fn != fn) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) fn, (int) threads, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        per_cu *= 2;
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

u32 exact_lds_bytes_host(u32 cap, u32 rawwords, u32 wordbytes) { return exact_lds_bytes(cap, rawwords, wordbytes); }

hipError_t launch_exact(hipStream_t st, u32 n_cu, u64 n_units, const DevParams &P, const DevBatch &B, const DevTable &T,
                        const u32 *wl, u32 *wl_count, u32 *wl_count_next, u32 wl_cap, const SegResults &R, u32 cap, u32 rawwords,
                        u32 max_seg_len) {
    // lane_bounds needs every staged segment (< cap) to fit its NW words
    const bool wide = P.max_mer > 32;  // 128-bit words, k_mer_check_128 (kmer.cpp:100, 180)
    const u32 lds = exact_lds_bytes(cap, rawwords, wide ? 16u : 8u);
    // max_seg_len = longest segment decide() is ever called on (halves, whole-read check, slices)
    // (k = 64 itself has no lane bound -- decide() walks its windows -- but every smaller k of a MAX_MER = 64 run does)
    const int nw = (P.flags & TREW_FLAG_NO_FILTER) ? 0 : (max_seg_len <= 95 ? 3 : max_seg_len <= 159 ? 5 : max_seg_len <= 319 ? 10 : 0);
    // one block = one wave; fill the chip exactly once (persistent, self-scheduling waves)
    typedef void (*kern_t)(DevParams, DevBatch, DevTable, const u32 *, u32 *, u32 *, u32, SegResults, u32, u32);
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
    // Self-scheduling waves: any grid size is correct, it only has to be large enough to keep the chip busy.  A big
    // batch gets every resident wave slot; a small one (the CLI's ~10^5-read batches, of which 1-2 % survive the
    // prefilter) one wave per 16 units, so that a launch does not start thousands of waves that find the queue empty.
    const u64 full = (u64) n_cu * (u64) per_cu;
    const u32 grid = (u32) std::min<u64>(full, std::max<u64>(n_units / 16, 64));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, st, P, B, T, wl, wl_count, wl_count_next, wl_cap, R, cap, rawwords);
    return hipGetLastError();
}

hipError_t launch_add_rows(hipStream_t st, const DevTable &T, const trew_hip_row *d_rows, u64 n) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(table_add_rows_kernel, dim3((u32) ((n + 255) / 256)), dim3(256), 0, st, T, d_rows, n);
    return hipGetLastError();
}

hipError_t launch_compact(hipStream_t st, const DevTable &T, u64 n_slots, u32 wide_log2_slots, int table, trew_hip_row *d_rows, u64 cap,
                          unsigned long long *d_n) {
    const u64 total = n_slots + (1ull << wide_log2_slots);
    hipLaunchKernelGGL(table_compact_kernel, dim3((u32) ((total + 255) / 256)), dim3(256), 0, st, T, n_slots, table, d_rows, cap, d_n);
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
