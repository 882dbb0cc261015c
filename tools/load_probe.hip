// load_probe.hip -- how long does ONE workgroup per CU (1024 threads, 140 KiB LDS) wait for a 128 KiB polynomial?
// Variants of the request shape; steady state over many rounds; optional 128 KiB write-back per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/load_probe.hip -o tools/load_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE, bool STORE, int SPIN>
__global__ __launch_bounds__(1024) void k(const double* x, double* y, unsigned long long* cyc) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const double* px = x + size_t(blockIdx.x) * 16384;
    double* py = y + size_t(blockIdx.x) * 16384;
    const unsigned long long t0 = __builtin_readcyclecounter();
    double v[16];
    if (MODE == 0) {                                  // production: 16 x 8 B per thread, rows of 1024
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = px[r * 1024 + tid];
    } else if (MODE == 1) {                           // 8 x 16 B per thread, linear
#pragma unroll
        for (int r = 0; r < 8; ++r) { d2 t = *reinterpret_cast<const d2*>(px + (r * 1024 + tid) * 2); v[2 * r] = t[0]; v[2 * r + 1] = t[1]; }
    } else if (MODE == 2) {                           // each wave reads its own contiguous 8 KiB (16 x 512 B)
        const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = px[wv * 1024 + r * 64 + ln];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[r]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    // stand-in for the transform's compute time
    for (int i = 0; i < SPIN; ++i) __builtin_amdgcn_s_sleep(16);
    double s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    if (STORE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) py[r * 1024 + tid] = v[r] + s;
    } else if (s == 1.2345) py[tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
    if (tid == 0) lds[0] = s;
}
template <int MODE, bool STORE, int SPIN>
void run(const char* name, const double* x, double* y, unsigned long long* cyc, int batch) {
    auto kern = k<MODE, STORE, SPIN>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 143360);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(batch), dim3(1024), 143360, 0, x, y, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(batch), dim3(1024), 143360, 0, x, y, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(size_t(batch) * 16); hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto q : hc) c += double(q);
    printf("%-58s %7.3f ms  %6.2f us/slot  input wait %6.0f cycles/wave  (%.2f TB/s read)\n", name, ms, ms * 1e3 / (batch / 256.0), double(c) / (double(batch) * 16), batch * 131072.0 / ms / 1e9);
}
int main() {
    const int batch = 8192;
    double *x, *y; unsigned long long* cyc;
    hipMalloc(&x, size_t(batch) * 131072); hipMalloc(&y, size_t(batch) * 131072); hipMalloc(&cyc, size_t(batch) * 16 * 8);
    hipMemset(x, 0, size_t(batch) * 131072);
    run<0, false, 0>("16 x 8 B rows, no store, no compute", x, y, cyc, batch);
    run<1, false, 0>("8 x 16 B linear, no store, no compute", x, y, cyc, batch);
    run<2, false, 0>("wave-contiguous 8 KiB, no store, no compute", x, y, cyc, batch);
    run<0, true, 0>("16 x 8 B rows, store, no compute", x, y, cyc, batch);
    run<1, true, 0>("8 x 16 B linear, store, no compute", x, y, cyc, batch);
    run<0, true, 30>("16 x 8 B rows, store, ~15 us of sleep", x, y, cyc, batch);
    run<1, true, 30>("8 x 16 B linear, store, ~15 us of sleep", x, y, cyc, batch);
    run<2, true, 30>("wave-contiguous, store, ~15 us of sleep", x, y, cyc, batch);
    run<0, false, 30>("16 x 8 B rows, no store, ~15 us of sleep", x, y, cyc, batch);
    return 0;
}
