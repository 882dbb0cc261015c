// atomic_bench.hip -- throughput of coalesced no-return FP64 atomic adds to global memory on gfx950 (would a
// mod-up kernel that accumulates its key products straight into prod beat materialising u?).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o tools/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* dst, const double* src, size_t n, int reps) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = src[i];
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) unsafeAtomicAdd(&dst[(i + size_t(r) * 0) % n], v + r);       // same target each rep
        if (MODE == 1) dst[i] = v + r;                                                 // plain store for reference
        if (MODE == 2) unsafeAtomicAdd(&dst[i ^ (size_t(r) << 20)], v + r);          // different lines each rep
    }
}
template <int MODE> void run(const char* name, double* d, const double* s, size_t n, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, s, n, reps); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, s, n, reps); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.2f G ops/s  (%.2f TB/s of 8-byte operands)\n", name, ms, double(n) * reps / ms / 1e6, double(n) * reps * 8 / ms / 1e9);
}
int main() {
    const size_t n = size_t(1) << 27;   // 1 GiB of doubles
    double *d, *s; hipMalloc(&d, n * 8); hipMalloc(&s, n * 8); hipMemset(d, 0, n * 8); hipMemset(s, 0, n * 8);
    run<1>("plain stores, 1 per element", d, s, n, 1);
    run<0>("atomic add f64, 1 per element", d, s, n, 1);
    run<0>("atomic add f64, 7 per element (same line)", d, s, n, 7);
    run<2>("atomic add f64, 7 per element (spread)", d, s, n, 7);
    return 0;
}
