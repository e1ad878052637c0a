// Micro-benchmark: issue rate of the integer VALU ops the TREW kernels are made of (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box.
// Every op is emitted through inline asm so that the compiler cannot fold the chains (an earlier
// version of this tool let LLVM simplify the xor chain and reported an impossible 1.5 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 9, a6 = a0 * 11, a7 = ~a0;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %1, %2, %1\n v_xor_b32 %2, %3, %2\n v_xor_b32 %3, %4, %3\n v_xor_b32 %4, %5, %4\n v_xor_b32 %5, %6, %5\n v_xor_b32 %6, %7, %6\n v_xor_b32 %7, %0, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 1) { REP8(asm volatile("v_alignbit_b32 %0, %1, %0, %8\n v_alignbit_b32 %1, %2, %1, %8\n v_alignbit_b32 %2, %3, %2, %8\n v_alignbit_b32 %3, %4, %3, %8\n v_alignbit_b32 %4, %5, %4, %8\n v_alignbit_b32 %5, %6, %5, %8\n v_alignbit_b32 %6, %7, %6, %8\n v_alignbit_b32 %7, %0, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
        if (OP == 2) { REP8(asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %3, %4, %3\n v_bcnt_u32_b32 %4, %5, %4\n v_bcnt_u32_b32 %5, %6, %5\n v_bcnt_u32_b32 %6, %7, %6\n v_bcnt_u32_b32 %7, %0, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 3) { REP8(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x48\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x48\n v_bitop3_b32 %2, %2, %3, %4 bitop3:0x48\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x48\n v_bitop3_b32 %4, %4, %5, %6 bitop3:0x48\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x48\n v_bitop3_b32 %6, %6, %7, %0 bitop3:0x48\n v_bitop3_b32 %7, %7, %0, %1 bitop3:0x48" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_add_u32 %0, %1, %0\n v_add_u32 %1, %2, %1\n v_add_u32 %2, %3, %2\n v_add_u32 %3, %4, %3\n v_add_u32 %4, %5, %4\n v_add_u32 %5, %6, %5\n v_add_u32 %6, %7, %6\n v_add_u32 %7, %0, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 8192 * 4);
    const char *names[5] = {"v_xor_b32 (VOP2)", "v_alignbit_b32 (VOP3)", "v_bcnt_u32_b32 (VOP3)", "v_bitop3_b32 (VOP3)", "v_add_u32 (VOP2)"};
    for (int op = 0; op < 5; op++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000, blocks = 256 * 8;  // 8 blocks of 256 per CU = 8 waves/SIMD
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            if (op == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double) blocks * 4 * iters * 64;  // wave-instructions: 4 waves/block x 64 ops per iteration
        double per_simd_cycle = winstr / (ms * 1e-3 * 2.4e9 * 1024);
        printf("%-24s %.3f ms  %.3f wave-instr/cycle/SIMD (at 2.4 GHz)  => %.2f cycles per wave-instr\n", names[op], ms, per_simd_cycle, 1.0 / per_simd_cycle);
    }
    return 0;
}
