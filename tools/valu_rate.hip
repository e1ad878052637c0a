// Micro-benchmark: issue rate of the integer VALU ops the TREW kernels are made of (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 9, a6 = a0 * 11, a7 = ~a0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) { a0 ^= a1; a1 ^= a2; a2 ^= a3; a3 ^= a4; a4 ^= a5; a5 ^= a6; a6 ^= a7; a7 ^= a0; }
            if (OP == 1) { a0 = __builtin_amdgcn_alignbit(a1, a0, s); a1 = __builtin_amdgcn_alignbit(a2, a1, s); a2 = __builtin_amdgcn_alignbit(a3, a2, s); a3 = __builtin_amdgcn_alignbit(a4, a3, s);
                           a4 = __builtin_amdgcn_alignbit(a5, a4, s); a5 = __builtin_amdgcn_alignbit(a6, a5, s); a6 = __builtin_amdgcn_alignbit(a7, a6, s); a7 = __builtin_amdgcn_alignbit(a0, a7, s); }
            if (OP == 2) { a0 = __popc(a1) + a0; a1 = __popc(a2) + a1; a2 = __popc(a3) + a2; a3 = __popc(a4) + a3; a4 = __popc(a5) + a4; a5 = __popc(a6) + a5; a6 = __popc(a7) + a6; a7 = __popc(a0) + a7; }
            if (OP == 3) { a0 = (a0 & a1) ^ a2; a1 = (a1 & a2) ^ a3; a2 = (a2 & a3) ^ a4; a3 = (a3 & a4) ^ a5; a4 = (a4 & a5) ^ a6; a5 = (a5 & a6) ^ a7; a6 = (a6 & a7) ^ a0; a7 = (a7 & a0) ^ a1; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 8192 * 4);
    const char *names[4] = {"v_xor (VOP2)", "v_alignbit (VOP3)", "v_bcnt (VOP3, acc)", "and+xor -> bitop3"};
    for (int op = 0; op < 4; op++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000, blocks = 256 * 8;  // 8 blocks of 256 per CU = 8 waves/SIMD
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double) blocks * 4 * iters * 64;  // wave-instructions (8 ops x 8 unroll per iter)
        double per_simd_cycle = winstr / (ms * 1e-3 * 2.4e9 * 1024);
        printf("%-22s %.3f ms  %.3f wave-instr/cycle/SIMD (at 2.4 GHz)  => %.2f cycles per wave-instr\n", names[op], ms, per_simd_cycle, 1.0 / per_simd_cycle);
    }
    return 0;
}
