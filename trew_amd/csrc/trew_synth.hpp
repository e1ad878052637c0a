// trew_synth.hpp -- counter-based synthetic read generator.
//
// Every base of every read is a pure function of (seed, read index, position),
// so the host (ASCII, for the oracle) and the device (packed triples, for the
// bench) regenerate bit-identical reads without sharing state.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TREW_SYNTH_HD __host__ __device__
#else
#define TREW_SYNTH_HD
#endif

namespace trew_synth {

// splitmix64 finaliser
TREW_SYNTH_HD inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
TREW_SYNTH_HD inline uint64_t key(uint64_t seed, uint64_t read, uint64_t pos, uint64_t stream) {
    return mix64(mix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) + read * 0xA24BAED4963EE407ull + pos * 0x9FB21C651E98DF25ull);
}

// codes: T=0 G=1 C=2 A=3, complement = 3 - c.  TTAGGG = {0,0,3,1,1,1}
TREW_SYNTH_HD inline int motif_base(uint32_t q) {
    const uint32_t packed = (0u) | (0u << 2) | (3u << 4) | (1u << 6) | (1u << 8) | (1u << 10);
    return (int) ((packed >> (2 * (q % 6))) & 3);
}

enum { KIND_RANDOM = 0, KIND_TELO = 1, KIND_JUNCTION = 2 };

struct ReadClass {
    int kind;
    uint32_t phase;
    int rc;
};

// 1.0 % fully telomeric, 0.5 % junction (telomeric first half, random second half),
// both reverse-complemented with p = 0.5.
TREW_SYNTH_HD inline ReadClass read_class(uint64_t seed, uint64_t read) {
    uint64_t h = key(seed, read, 0, 1);
    uint32_t u = (uint32_t) (h % 100000u);
    ReadClass c;
    c.kind = u < 1000u ? KIND_TELO : (u < 1500u ? KIND_JUNCTION : KIND_RANDOM);
    c.phase = (uint32_t) ((h >> 20) % 6u);
    c.rc = (int) ((h >> 40) & 1u);
    return c;
}

// base `pos` (0-based) of a short read of length L; returns 0..3 or 4 for N
TREW_SYNTH_HD inline int short_base(uint64_t seed, uint64_t read, const ReadClass &c, uint32_t pos, uint32_t L) {
    // position in the un-reversed read
    uint32_t q = c.rc ? (L - 1 - pos) : pos;
    int b;
    bool repeat_part = c.kind == KIND_TELO || (c.kind == KIND_JUNCTION && q < L / 2);
    if (repeat_part) {
        b = motif_base(q + c.phase);
        uint64_t e = key(seed, read, q, 3);
        if ((uint32_t) (e % 10000u) < 100u) b = (b + 1 + (int) ((e >> 32) % 3u)) & 3;  // 1 % substitutions
    } else {
        b = (int) (key(seed, read, q, 2) & 3u);
    }
    if (c.rc) b = 3 - b;
    uint64_t nn = key(seed, read, pos, 4);
    if ((uint32_t) (nn % 1000000u) < 500u) return 4;  // N with p = 5e-4
    return b;
}

// ---- paired fragments (config 3): fragment of 2L bases; R1 = first L, R2 = revcomp(last L).
// 1 % fully telomeric fragments, 0.5 % telomeric in the R1 half only; whole
// fragment reverse-complemented with p = 0.5 (which swaps the roles of the mates).
TREW_SYNTH_HD inline int frag_base(uint64_t seed, uint64_t frag, const ReadClass &c, uint32_t fpos, uint32_t FL) {
    uint32_t q = c.rc ? (FL - 1 - fpos) : fpos;
    int b;
    bool repeat_part = c.kind == KIND_TELO || (c.kind == KIND_JUNCTION && q < FL / 2);
    if (repeat_part) {
        b = motif_base(q + c.phase);
        uint64_t e = key(seed, frag, q, 3);
        if ((uint32_t) (e % 10000u) < 100u) b = (b + 1 + (int) ((e >> 32) % 3u)) & 3;
    } else {
        b = (int) (key(seed, frag, q, 2) & 3u);
    }
    if (c.rc) b = 3 - b;
    return b;
}
// mate 0: R1[pos] = frag[pos]; mate 1: R2[pos] = comp(frag[2L-1-pos])
TREW_SYNTH_HD inline int pair_base(uint64_t seed, uint64_t frag, const ReadClass &c, int mate, uint32_t pos, uint32_t L) {
    int b = mate == 0 ? frag_base(seed, frag, c, pos, 2 * L) : 3 - frag_base(seed, frag, c, 2 * L - 1 - pos, 2 * L);
    uint64_t nn = key(seed, frag * 2 + (uint64_t) mate, pos, 4);
    if ((uint32_t) (nn % 1000000u) < 500u) return 4;
    return b;
}

// ---- long reads (config 4): length drawn from a 1024-entry quantile table of
// clip(lognormal(mu = 9.413, sigma = 0.7), 1000, 200000) (median ~12.2 kb, N50 ~20 kb); 5 % carry a
// 2-6 kb (TTAGGG)n tail at the 3' end with 5 % substitutions (ONT-like); half of all reads are
// reverse-complemented.  The table is computed once on the host (long_quantiles) and shared with the
// device, so lengths never depend on device math.
constexpr int kLongQuantiles = 1024;

struct LongClass {
    uint32_t len, tail_len, phase;
    int rc;
};

TREW_SYNTH_HD inline LongClass long_class(uint64_t seed, uint64_t read, const uint32_t *qtable) {
    const uint64_t h = key(seed, read, 0, 5);
    LongClass c;
    c.len = qtable[h % (uint64_t) kLongQuantiles];
    const uint32_t u = (uint32_t) ((h >> 12) % 100u);
    c.tail_len = u < 5u ? 2000u + (uint32_t) ((h >> 24) % 4001u) : 0u;
    if (c.tail_len > c.len) c.tail_len = c.len;
    c.phase = (uint32_t) ((h >> 40) % 6u);
    c.rc = (int) ((h >> 50) & 1u);
    return c;
}

TREW_SYNTH_HD inline int long_base(uint64_t seed, uint64_t read, const LongClass &c, uint32_t pos) {
    const uint32_t q = c.rc ? (c.len - 1 - pos) : pos;
    int b;
    if (c.tail_len && q >= c.len - c.tail_len) {
        b = motif_base(q + c.phase);
        const uint64_t e = key(seed, read, q, 6);
        if ((uint32_t) (e % 100u) < 5u) b = (b + 1 + (int) ((e >> 32) % 3u)) & 3;  // 5 % substitutions
    } else {
        b = (int) (key(seed, read, q, 7) & 3u);
    }
    return c.rc ? 3 - b : b;
}

TREW_SYNTH_HD inline char base_char(int b) { return b == 0 ? 'T' : b == 1 ? 'G' : b == 2 ? 'C' : b == 3 ? 'A' : 'N'; }

}  // namespace trew_synth
