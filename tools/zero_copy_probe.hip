// zero_copy_probe.hip -- what does it cost a kernel to read its input straight from pinned HOST memory / write its output there,
// against hipMemcpyAsync (SDMA) of the same bytes plus a kernel on device memory? Shapes of the lone-keyswitch path (keyswitch_lat.hip):
// k_ksq_intt reads 0.8 MB (24 workgroups x 256 threads, 16 words per thread, 2 KiB per wave instruction), k_ksq_down writes 1.6 MB
// (48 workgroups). Prints microseconds per variant (median of 200).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const u64* __restrict__ src, u64* __restrict__ dst) {     // 4096 words per workgroup
    const u64* s = src + size_t(blockIdx.x) * 4096;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = s[r * 256 + threadIdx.x];
    u64 acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += v[r] * (r + 1);
    dst[size_t(blockIdx.x) * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_write(u64* __restrict__ dst, u64 seed) {
    u64* d = dst + size_t(blockIdx.x) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r * 256 + threadIdx.x] = seed + r * 256 + threadIdx.x;
}
__global__ __launch_bounds__(256) void k_rmw(u64* __restrict__ dst, const u64* __restrict__ old, u64 seed) {
    u64* d = dst + size_t(blockIdx.x) * 4096;
    const u64* o = old + size_t(blockIdx.x) * 4096;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = o[r * 256 + threadIdx.x];
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r * 256 + threadIdx.x] = v[r] + seed;
}

template <class F>
static double median_us(hipStream_t st, F f, int reps = 200) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int i = 0; i < reps + 5; ++i) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        f();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 5) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t nin = 24 * 4096, nout = 48 * 4096;                // 0.79 MB in, 1.57 MB out
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    u64 *h_in, *h_out, *h_old, *d_in, *d_out, *d_tmp;
    CK(hipHostMalloc((void**)&h_in, nin * 8, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_out, nout * 8, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_old, nout * 8, hipHostMallocDefault));
    CK(hipMalloc((void**)&d_in, nin * 8)); CK(hipMalloc((void**)&d_out, nout * 8)); CK(hipMalloc((void**)&d_tmp, nout * 8));
    for (size_t i = 0; i < nin; ++i) h_in[i] = i * 3;
    for (size_t i = 0; i < nout; ++i) h_old[i] = i;
    u64 *dh_in, *dh_out, *dh_old;
    CK(hipHostGetDevicePointer((void**)&dh_in, h_in, 0)); CK(hipHostGetDevicePointer((void**)&dh_out, h_out, 0)); CK(hipHostGetDevicePointer((void**)&dh_old, h_old, 0));
    printf("device pointers of the pinned slabs %s the host pointers\n", (dh_in == h_in && dh_out == h_out) ? "ARE" : "are NOT");
    printf("H2D copy 0.79 MB (hipMemcpyAsync)            : %7.1f us\n", median_us(st, [&] { CK(hipMemcpyAsync(d_in, h_in, nin * 8, hipMemcpyHostToDevice, st)); }));
    printf("H2D copy + kernel reading device memory      : %7.1f us\n", median_us(st, [&] { CK(hipMemcpyAsync(d_in, h_in, nin * 8, hipMemcpyHostToDevice, st)); hipLaunchKernelGGL(k_read, dim3(24), dim3(256), 0, st, d_in, d_tmp); }));
    printf("kernel reading 0.79 MB of pinned host memory : %7.1f us\n", median_us(st, [&] { hipLaunchKernelGGL(k_read, dim3(24), dim3(256), 0, st, dh_in, d_tmp); }));
    printf("kernel reading the same from device memory   : %7.1f us\n", median_us(st, [&] { hipLaunchKernelGGL(k_read, dim3(24), dim3(256), 0, st, d_in, d_tmp); }));
    printf("D2H copy 1.57 MB (hipMemcpyAsync)            : %7.1f us\n", median_us(st, [&] { CK(hipMemcpyAsync(h_out, d_out, nout * 8, hipMemcpyDeviceToHost, st)); }));
    printf("kernel writing device memory + D2H copy      : %7.1f us\n", median_us(st, [&] { hipLaunchKernelGGL(k_write, dim3(48), dim3(256), 0, st, d_out, 7ull); CK(hipMemcpyAsync(h_out, d_out, nout * 8, hipMemcpyDeviceToHost, st)); }));
    printf("kernel writing 1.57 MB of pinned host memory : %7.1f us\n", median_us(st, [&] { hipLaunchKernelGGL(k_write, dim3(48), dim3(256), 0, st, dh_out, 7ull); }));
    bool ok = true; for (size_t i = 0; i < nout; ++i) ok = ok && h_out[i] == 7ull + (i % 4096);
    printf("   (host sees the kernel's words after the stream synchronisation: %s)\n", ok ? "yes" : "NO");
    printf("kernel: read 1.57 MB pinned + write 1.57 pinned: %7.1f us\n", median_us(st, [&] { hipLaunchKernelGGL(k_rmw, dim3(48), dim3(256), 0, st, dh_out, dh_old, 5ull); }));
    printf("H2D 0.79 + H2D 1.57 + kernel + D2H 1.57 (SDMA): %7.1f us\n", median_us(st, [&] {
        CK(hipMemcpyAsync(d_in, h_in, nin * 8, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(d_tmp, h_old, nout * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_rmw, dim3(48), dim3(256), 0, st, d_out, d_tmp, 5ull); CK(hipMemcpyAsync(h_out, d_out, nout * 8, hipMemcpyDeviceToHost, st)); }));
    return 0;
}
