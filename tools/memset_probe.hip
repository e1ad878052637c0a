// tools/memset_probe.hip -- ONE deterministic probe for the empty-tables flake of round 3 (profiles/r04/README.md):
// does hipMemset of device memory on the null stream return to the host before the fill has run, and does a kernel on a
// NON-BLOCKING stream launched right behind it start before the fill is complete?
//   hipcc --offload-arch=gfx950 -O2 -o tools/memset_probe tools/memset_probe.hip && tools/memset_probe
// Prints, for a 256 MiB buffer (the CLI's count table): host time of the hipMemset call, device time of the fill (events on the
// null stream around it), whether an event recorded behind the memset was already complete when the call returned, and how
// many words a kernel on a non-blocking stream, launched immediately after the call returned, still found unfilled.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                       \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__global__ void count_nonzero(const unsigned long long *p, size_t n, unsigned long long *out) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) c += p[i] != 0ull;
    if (c) atomicAdd(out, c);
}
__global__ void fill(unsigned long long *p, size_t n, unsigned long long v) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v;
}

int main() {
    const size_t bytes = 256ull << 20, n = bytes / 8;
    unsigned long long *d = nullptr, *d_out = nullptr, *h_out = nullptr;
    CHK(hipMalloc((void **) &d, bytes));
    CHK(hipMalloc((void **) &d_out, 8));
    CHK(hipHostMalloc((void **) &h_out, 8, hipHostMallocDefault));
    hipStream_t nb;
    CHK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipEventCreate(&e2));
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, nb, d, n, 0xffffffffffffffffull);  // every word non-zero
        CHK(hipMemsetAsync(d_out, 0, 8, nb));
        CHK(hipStreamSynchronize(nb));
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, nullptr));
        const auto t0 = std::chrono::steady_clock::now();
        CHK(hipMemset(d, 0, bytes));  // the call trew_hip_init / trew_hip_reset_tables make
        const auto t1 = std::chrono::steady_clock::now();
        CHK(hipEventRecord(e1, nullptr));
        const hipError_t q = hipEventQuery(e1);  // already complete when the call has returned?
        // what a first batch submitted right away would have seen: a kernel on a non-blocking stream, no synchronisation
        hipLaunchKernelGGL(count_nonzero, dim3(4096), dim3(256), 0, nb, d, n, d_out);
        CHK(hipMemcpyAsync(h_out, d_out, 8, hipMemcpyDeviceToHost, nb));
        CHK(hipStreamSynchronize(nb));
        CHK(hipEventSynchronize(e1));
        float fill_ms = 0;
        CHK(hipEventElapsedTime(&fill_ms, e0, e1));
        const double host_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        printf("rep %d: hipMemset(256 MiB) host call %.3f ms, null-stream events around it %.3f ms, event behind it %s at return, "
               "kernel on a non-blocking stream right behind the call found %llu of %zu words still unfilled\n",
               rep, host_ms, fill_ms, q == hipSuccess ? "COMPLETE" : "NOT complete", *h_out, n);
    }
    return 0;
}
