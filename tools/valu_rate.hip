// Micro-benchmark: issue rate of the integer VALU ops the TREW kernels are made of (gfx950), and the clock the chip holds
// while running them.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box:
//   tools/valu_rate [waves_per_simd = 8] [json]
// Every op is emitted through inline asm so that the compiler cannot fold the chains (an earlier version of this tool let
// LLVM simplify the xor chain and reported an impossible 1.5 cycles).  Eight independent chains per wave, `waves_per_simd`
// waves on every SIMD.  The figure reported is SIMD cycles per wave-instruction = kernel time x clock / instructions per SIMD,
// with the clock the chip actually held during that kernel: delta s_memtime / delta s_memrealtime x 100 MHz, median over
// waves (MI355X_MICROARCH.md, DVFS item 6) -- 1.9 to 2.4 GHz depending on the op.  (The duration of the MEDIAN wave is
// printed too but is not a throughput: issue arbitration favours older waves, which finish at 64 % of the kernel's time.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP8(X) X X X X X X X X
// one "round" = 8 instructions, one per chain; OPS2(a, b) is the text of `a = op(a, b)`
#define CHAIN8(OPS2)                                                                                                                   \
    asm volatile(OPS2("%0", "%1") OPS2("%1", "%2") OPS2("%2", "%3") OPS2("%3", "%4") OPS2("%4", "%5") OPS2("%5", "%6") OPS2("%6", "%7") \
                     OPS2("%7", "%0")                                                                                                  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                                      \
                 : "s"(s))
#define CHAIN8_64(OPS2)                                                                                                                \
    asm volatile(OPS2("%0", "%1") OPS2("%1", "%2") OPS2("%2", "%3") OPS2("%3", "%4") OPS2("%4", "%5") OPS2("%5", "%6") OPS2("%6", "%7") \
                     OPS2("%7", "%0")                                                                                                  \
                 : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)                                      \
                 : "s"(s))

#define OP_XOR(a, b) "v_xor_b32 " a ", " b ", " a "\n"
#define OP_ALIGNBIT(a, b) "v_alignbit_b32 " a ", " b ", " a ", %8\n"
#define OP_BCNT(a, b) "v_bcnt_u32_b32 " a ", " b ", " a "\n"
#define OP_BITOP3(a, b) "v_bitop3_b32 " a ", " a ", " b ", " b " bitop3:0x48\n"
#define OP_ADD(a, b) "v_add_u32 " a ", " b ", " a "\n"
#define OP_LSHR32(a, b) "v_lshrrev_b32 " a ", %8, " a "\n"
#define OP_LSHLOR(a, b) "v_lshl_or_b32 " a ", " b ", %8, " a "\n"
#define OP_MAX3(a, b) "v_max3_u32 " a ", " a ", " b ", " b "\n"
#define OP_SUB(a, b) "v_sub_u32 " a ", " b ", " a "\n"
#define OP_AND(a, b) "v_and_b32 " a ", " b ", " a "\n"
#define OP_BFE(a, b) "v_bfe_u32 " a ", " b ", %8, 5\n"
#define OP_PERM(a, b) "v_perm_b32 " a ", " a ", " b ", " b "\n"
#define OP_CMP(a, b) "v_cmp_le_u32 vcc, " a ", " b "\n"
#define OP_MOV(a, b) "v_mov_b32 " a ", " b "\n"
#define OP_LSHR64(a, b) "v_lshrrev_b64 " a ", %8, " b "\n"
#define OP_LSHL64(a, b) "v_lshlrev_b64 " a ", %8, " b "\n"
#define OP_MADU64(a, b) "v_mad_u64_u32 " a ", vcc, " b ", " b ", " a "\n"
#define OP_ADDC(a, b) "v_add_co_u32 " a ", vcc, " b ", " a "\n"
#define OP_CNDMASK(a, b) "v_cndmask_b32 " a ", " a ", " b ", vcc\n"
#define OP_MBCNT(a, b) "v_mbcnt_lo_u32_b32 " a ", " b ", " a "\n"
#define OP_FFBL(a, b) "v_ffbl_b32 " a ", " b "\n"
#define OP_BFREV(a, b) "v_bfrev_b32 " a ", " b "\n"
#define OP_MULLO(a, b) "v_mul_lo_u32 " a ", " b ", " a "\n"
#define OP_SAD(a, b) "v_sad_u8 " a ", " a ", " b ", " a "\n"
#define OP_DOT4(a, b) "v_dot4_u32_u8 " a ", " a ", " b ", " a "\n"
// round 4: the cross-lane and select ops the exact kernel is full of (scalar destinations s20..s27 are clobbered)
#define OP_MAXU(a, b) "v_max_u32 " a ", " b ", " a "\n"
#define OP_MINU(a, b) "v_min_u32 " a ", " b ", " a "\n"
#define OP_OR(a, b) "v_or_b32 " a ", " b ", " a "\n"
#define OP_LSHL32(a, b) "v_lshlrev_b32 " a ", %8, " a "\n"
#define OP_MUL24(a, b) "v_mul_u32_u24 " a ", " b ", " a "\n"
#define OP_DPPMOV(a, b) "v_mov_b32_dpp " a ", " b " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_DPPOR(a, b) "v_or_b32_dpp " a ", " b ", " a " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define OP_WRITELANE(a, b) "v_writelane_b32 " a ", %8, 3\n"
#define OP_CMPEQ(a, b) "v_cmp_eq_u32 vcc, " a ", " b "\n"
#define OP_CMP64(a, b) "v_cmp_lt_u64 vcc, " a ", " b "\n"

// round 4, second batch: three-operand adds / logic and packed 16-bit arithmetic (candidates for the prefilter's threshold tail)
#define OP_ADD3(a, b) "v_add3_u32 " a ", " a ", " b ", " b "\n"
#define OP_ANDOR(a, b) "v_and_or_b32 " a ", " a ", " b ", " b "\n"
#define OP_OR3(a, b) "v_or3_b32 " a ", " a ", " b ", " b "\n"
#define OP_PKADD(a, b) "v_pk_add_u16 " a ", " b ", " a "\n"
#define OP_PKMAX(a, b) "v_pk_max_u16 " a ", " b ", " a "\n"
#define OP_PKSUB(a, b) "v_pk_sub_u16 " a ", " b ", " a "\n"
#define OP_SUBS(a, b) "v_sub_u32 " a ", %8, " b "\n"
#define OP_CMPGTI(a, b) "v_cmp_gt_i32 vcc, 0, " a "\n"
#define OP_LSHLADD(a, b) "v_lshl_add_u32 " a ", " b ", 16, " a "\n"

// does a scalar source operand cost issue time?
#define OP_ANDS(a, b) "v_and_b32 " a ", %8, " b "\n"
#define OP_BITOP3S(a, b) "v_bitop3_b32 " a ", " a ", " b ", %8 bitop3:0x28\n"
#define OP_BCNTS(a, b) "v_bcnt_u32_b32 " a ", " b ", %8\n"
#define OP_XORS(a, b) "v_xor_b32 " a ", %8, " b "\n"

#define OP_XORLIT(a, b) "v_xor_b32 " a ", 0x12345678, " b "\n"
#define OP_XORINL(a, b) "v_xor_b32 " a ", 15, " b "\n"
#define OP_ADDINL(a, b) "v_add_u32 " a ", 8, " b "\n"
#define OP_XORVCC(a, b) "v_xor_b32 " a ", vcc_lo, " b "\n"

enum { N_OPS = 55 };
static const char *kNames[N_OPS] = {"v_xor_b32",     "v_alignbit_b32", "v_bcnt_u32_b32", "v_bitop3_b32",  "v_add_u32",      "v_lshrrev_b32", "v_lshl_or_b32",
                                    "v_max3_u32",    "v_sub_u32",      "v_and_b32",      "v_bfe_u32",     "v_perm_b32",     "v_cmp_le_u32",  "v_mov_b32",
                                    "v_lshrrev_b64", "v_lshlrev_b64",  "v_mad_u64_u32",  "v_add_co_u32",  "v_cndmask_b32",  "v_mbcnt_lo",    "v_ffbl_b32",
                                    "v_bfrev_b32",   "v_mul_lo_u32",   "v_sad_u8",       "v_dot4_u32_u8", "xor+alignbit+bcnt mix (the prefilter's k loop)",
                                    "v_max_u32",     "v_min_u32",      "v_or_b32",       "v_lshlrev_b32", "v_mul_u32_u24",  "v_mov_b32_dpp", "v_or_b32_dpp",
                                    "v_writelane_b32", "v_cmp_eq_u32", "v_cmp_lt_u64",   "v_readlane_b32", "v_readfirstlane_b32",
                                    "v_add3_u32",    "v_and_or_b32",   "v_or3_b32",      "v_pk_add_u16",  "v_pk_max_u16",   "v_pk_sub_u16",  "v_sub_u32 (sgpr)",
                                    "v_cmp_gt_i32",  "v_lshl_add_u32", "v_and_b32 (sgpr)", "v_bitop3_b32 (sgpr)", "v_bcnt_u32_b32 (sgpr)", "v_xor_b32 (sgpr)",
                                    "v_xor_b32 (literal)", "v_xor_b32 (inline constant)", "v_add_u32 (inline constant)", "v_xor_b32 (vcc_lo)"};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *stamps, int iters, unsigned s) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 9, a6 = a0 * 11, a7 = ~a0;
    unsigned long long b0 = a0 * 0x9E3779B97F4A7C15ull, b1 = b0 * 3, b2 = b0 * 5, b3 = b0 * 7, b4 = ~b0, b5 = b0 + 9, b6 = b0 * 11, b7 = b0 ^ 0x5555;
    asm volatile("s_mov_b64 s[20:21], 0x55555555" ::: "s20", "s21");  // select mask of the v_cndmask test
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(CHAIN8(OP_XOR);) }
        if (OP == 1) { REP8(CHAIN8(OP_ALIGNBIT);) }
        if (OP == 2) { REP8(CHAIN8(OP_BCNT);) }
        if (OP == 3) { REP8(CHAIN8(OP_BITOP3);) }
        if (OP == 4) { REP8(CHAIN8(OP_ADD);) }
        if (OP == 5) { REP8(CHAIN8(OP_LSHR32);) }
        if (OP == 6) { REP8(CHAIN8(OP_LSHLOR);) }
        if (OP == 7) { REP8(CHAIN8(OP_MAX3);) }
        if (OP == 8) { REP8(CHAIN8(OP_SUB);) }
        if (OP == 9) { REP8(CHAIN8(OP_AND);) }
        if (OP == 10) { REP8(CHAIN8(OP_BFE);) }
        if (OP == 11) { REP8(CHAIN8(OP_PERM);) }
        if (OP == 12) { REP8(CHAIN8(OP_CMP);) }
        if (OP == 13) { REP8(CHAIN8(OP_MOV);) }
        if (OP == 14) { REP8(CHAIN8_64(OP_LSHR64);) }
        if (OP == 15) { REP8(CHAIN8_64(OP_LSHL64);) }
        if (OP == 16) { REP8(asm volatile(OP_MADU64("%0", "%1") OP_MADU64("%0", "%2") OP_MADU64("%0", "%1") OP_MADU64("%0", "%2") OP_MADU64("%3", "%1") OP_MADU64("%3", "%2") OP_MADU64("%3", "%1") OP_MADU64("%3", "%2") : "+v"(b0), "+v"(a1), "+v"(a2), "+v"(b3) : : "vcc");) }
        if (OP == 17) { REP8(CHAIN8(OP_ADDC);) }
        if (OP == 18) {
            // the select mask in an SGPR pair set once (with vcc, which nothing here writes, the first version measured 22.8 cycles)
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n"
                              "v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]\n v_cndmask_b32_e64 %5, %5, %6, s[20:21]\n"
                              "v_cndmask_b32_e64 %6, %6, %7, s[20:21]\n v_cndmask_b32_e64 %7, %7, %0, s[20:21]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "s"(s)
                              : "s20", "s21");)
        }
        if (OP == 19) { REP8(CHAIN8(OP_MBCNT);) }
        if (OP == 20) { REP8(CHAIN8(OP_FFBL);) }
        if (OP == 21) { REP8(CHAIN8(OP_BFREV);) }
        if (OP == 22) { REP8(CHAIN8(OP_MULLO);) }
        if (OP == 23) { REP8(CHAIN8(OP_SAD);) }
        if (OP == 24) { REP8(CHAIN8(OP_DOT4);) }
        if (OP == 26) { REP8(CHAIN8(OP_MAXU);) }
        if (OP == 27) { REP8(CHAIN8(OP_MINU);) }
        if (OP == 28) { REP8(CHAIN8(OP_OR);) }
        if (OP == 29) { REP8(CHAIN8(OP_LSHL32);) }
        if (OP == 30) { REP8(CHAIN8(OP_MUL24);) }
        if (OP == 31) { REP8(CHAIN8(OP_DPPMOV);) }
        if (OP == 32) { REP8(CHAIN8(OP_DPPOR);) }
        if (OP == 33) { REP8(CHAIN8(OP_WRITELANE);) }
        if (OP == 34) { REP8(CHAIN8(OP_CMPEQ);) }
        if (OP == 35) { REP8(CHAIN8_64(OP_CMP64);) }
        if (OP == 36) {
            REP8(asm volatile("v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %1, 2\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 4\n"
                              "v_readlane_b32 s24, %4, 5\n v_readlane_b32 s25, %5, 6\n v_readlane_b32 s26, %6, 7\n v_readlane_b32 s27, %7, 8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "s"(s)
                              : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        }
        if (OP == 37) {
            REP8(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n"
                              "v_readfirstlane_b32 s24, %4\n v_readfirstlane_b32 s25, %5\n v_readfirstlane_b32 s26, %6\n v_readfirstlane_b32 s27, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "s"(s)
                              : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        }
        if (OP == 38) { REP8(CHAIN8(OP_ADD3);) }
        if (OP == 39) { REP8(CHAIN8(OP_ANDOR);) }
        if (OP == 40) { REP8(CHAIN8(OP_OR3);) }
        if (OP == 41) { REP8(CHAIN8(OP_PKADD);) }
        if (OP == 42) { REP8(CHAIN8(OP_PKMAX);) }
        if (OP == 43) { REP8(CHAIN8(OP_PKSUB);) }
        if (OP == 44) { REP8(CHAIN8(OP_SUBS);) }
        if (OP == 45) { REP8(CHAIN8(OP_CMPGTI);) }
        if (OP == 46) { REP8(CHAIN8(OP_LSHLADD);) }
        if (OP == 47) { REP8(CHAIN8(OP_ANDS);) }
        if (OP == 48) { REP8(CHAIN8(OP_BITOP3S);) }
        if (OP == 49) { REP8(CHAIN8(OP_BCNTS);) }
        if (OP == 50) { REP8(CHAIN8(OP_XORS);) }
        if (OP == 51) { REP8(CHAIN8(OP_XORLIT);) }
        if (OP == 52) { REP8(CHAIN8(OP_XORINL);) }
        if (OP == 53) { REP8(CHAIN8(OP_ADDINL);) }
        if (OP == 54) { REP8(CHAIN8(OP_XORVCC);) }
        if (OP == 25) {
            // the instruction mix of one (k, word) of the prefilter's fast path: 2 alignbit, 2 xor, 1 and, 3 bcnt
            REP8(asm volatile("v_alignbit_b32 %4, %1, %0, %8\n v_alignbit_b32 %5, %3, %2, %8\n v_xor_b32 %4, %4, %0\n v_xor_b32 %5, %5, %2\n"
                              "v_bcnt_u32_b32 %6, %4, %6\n v_bcnt_u32_b32 %7, %5, %7\n v_and_b32 %4, %4, %5\n v_bcnt_u32_b32 %6, %4, %6\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "s"(s));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned) (b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7);
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[4 * w] = t1 - t0;
        stamps[4 * w + 1] = r1 - r0;
        stamps[4 * w + 2] = r0;
        stamps[4 * w + 3] = r1;
    }
}

typedef void (*kern_t)(unsigned *, unsigned long long *, int, unsigned);
template <int OP>
struct Table {
    static void fill(kern_t *t) {
        t[OP] = k<OP>;
        Table<OP - 1>::fill(t);
    }
};
template <>
struct Table<-1> {
    static void fill(kern_t *) {}
};

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 8;  // waves per SIMD = 256-thread blocks per CU
    const bool json = argc > 2 && !strcmp(argv[2], "json");
    hipDeviceProp_t prop;
    (void) hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const int blocks = n_cu * wps, waves = blocks * 4;
    unsigned *d;
    unsigned long long *d_st;
    (void) hipMalloc(&d, (size_t) blocks * 256 * 4);
    (void) hipMalloc(&d_st, (size_t) waves * 32);
    kern_t tab[N_OPS];
    Table<N_OPS - 1>::fill(tab);
    std::vector<unsigned long long> st(4 * (size_t) waves);
    if (json) printf("{\"waves_per_simd\": %d, \"cycles_per_wave_inst\": {", wps);
    double clock_sum = 0;
    for (int op = 0; op < N_OPS; op++) {
        hipEvent_t e0, e1;
        (void) hipEventCreate(&e0);
        (void) hipEventCreate(&e1);
        const int iters = 3000;
        for (int rep = 0; rep < 2; rep++) {
            (void) hipEventRecord(e0);
            hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, d, d_st, iters, 7u);
            (void) hipEventRecord(e1);
            (void) hipEventSynchronize(e1);
        }
        float ms;
        (void) hipEventElapsedTime(&ms, e0, e1);
        (void) hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc(waves), clk(waves);
        for (int w = 0; w < waves; w++) {
            cyc[w] = (double) st[4 * w];
            clk[w] = (double) st[4 * w] / (double) st[4 * w + 1] * 100e6;
        }
        // when did the waves start and end (s_memrealtime, 100 MHz)?  All resident at once <=> every wave starts within a few
        // microseconds of the first one and the kernel lasts as long as one wave.
        unsigned long long r_first = ~0ull, r_last_start = 0, r_end = 0;
        for (int w = 0; w < waves; w++) {
            r_first = std::min(r_first, st[4 * w + 2]);
            r_last_start = std::max(r_last_start, st[4 * w + 2]);
            r_end = std::max(r_end, st[4 * w + 3]);
        }
        const double span_ms = (double) (r_end - r_first) / 100e3, start_spread_ms = (double) (r_last_start - r_first) / 100e3;
        std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());
        std::nth_element(clk.begin(), clk.begin() + waves / 2, clk.end());
        const double winstr_per_wave = (double) iters * 64;  // 8 rounds of 8 instructions per iteration
        // wps waves share one SIMD: SIMD cycles per wave-instruction = kernel cycles / instructions issued on the SIMD
        const double per_inst = (double) ms * 1e-3 * clk[waves / 2] / (winstr_per_wave * wps);
        clock_sum += clk[waves / 2];
        if (json)
            printf("%s\"%s\": %.2f", op ? ", " : "", kNames[op], per_inst);
        else
            printf("%-48s %.3f ms  %.2f SIMD cycles per wave-instruction  (in-kernel clock %.2f GHz; first wave start to last wave end %.3f ms, wave starts spread over %.3f ms, median wave %.3f ms)\n",
                   kNames[op], ms, per_inst, clk[waves / 2] * 1e-9, span_ms, start_spread_ms, cyc[waves / 2] / clk[waves / 2] * 1e3);
    }
    if (json) printf("}, \"clock_ghz\": %.3f, \"source\": \"tools/valu_rate.hip: kernel time x in-kernel clock / instructions per SIMD, 192 000 instructions per wave, %d waves per SIMD\"}\n", clock_sum / N_OPS * 1e-9, wps);
    return 0;
}
