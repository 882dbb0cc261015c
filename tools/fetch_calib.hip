// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the keyswitch kernels
// use (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves exactly 1 GiB (2^30 bytes) that is not resident in any cache (a 4 GiB buffer, a different quarter each).
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/fetch_calib     (and the same with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_read16(const ulonglong2* p, size_t n, unsigned long long* sink) {       // 16 B per lane, coalesced
    unsigned long long acc = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { ulonglong2 v = p[i]; acc += v.x ^ v.y; }
    if (acc == 0x1234567) *sink = acc;
}
__global__ void k_read8(const unsigned long long* p, size_t n, unsigned long long* sink) {  // 8 B per lane, coalesced (A order)
    unsigned long long acc = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc += p[i];
    if (acc == 0x1234567) *sink = acc;
}
__global__ void k_read8_strided(const unsigned long long* p, size_t n, unsigned long long* sink) {   // 8 B per lane, 32 B lane stride (B order, 4 passes per line)
    unsigned long long acc = 0;
    const size_t chunk = size_t(blockIdx.x) * blockDim.x * 4;                              // a block owns blockDim*4 consecutive words
    for (size_t base = chunk; base < n; base += size_t(gridDim.x) * blockDim.x * 4)
        for (int r = 0; r < 4; ++r) acc += p[base + size_t(threadIdx.x) * 4 + r];
    if (acc == 0x1234567) *sink = acc;
}
// the key stream's instruction: buffer_load_dwordx2 through a resource descriptor, the thread's byte offset in a VGPR, the ROW offset in an
// SGPR (RowStream, ntt_core.hpp), 8 B per lane, rows of 128 KiB walked 16 words per thread as mac_keys does
__global__ __launch_bounds__(1024) void k_read8_buffer_rows(const unsigned long long* p, size_t rows, unsigned long long* sink) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    unsigned long long acc = 0;
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const unsigned long long base = (unsigned long long)(p + row * 16384);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 131072, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(threadIdx.x * 8), r * 8192, 0);
            acc += x.x ^ x.y;
        }
    }
    if (acc == 0x1234567) *sink = acc;
}
__global__ void k_write8(unsigned long long* p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = i;
}
__global__ void k_write16(ulonglong2* p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = make_ulonglong2(i, i);
}

int main() {
    const size_t GiB = size_t(1) << 30;
    char* buf; unsigned long long* sink;
    hipMalloc(&buf, 6 * GiB); hipMalloc(&sink, 8);
    hipMemset(buf, 1, 6 * GiB);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const ulonglong2*)(buf + 0 * GiB), GiB / 16, sink);
    hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, (const unsigned long long*)(buf + 1 * GiB), GiB / 8, sink);
    hipLaunchKernelGGL(k_read8_strided, dim3(4096), dim3(256), 0, 0, (const unsigned long long*)(buf + 2 * GiB), GiB / 8, sink);
    hipLaunchKernelGGL(k_read8_buffer_rows, dim3(1024), dim3(1024), 0, 0, (const unsigned long long*)(buf + 5 * GiB), GiB / 131072, sink);
    hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, (unsigned long long*)(buf + 3 * GiB), GiB / 8);
    hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (ulonglong2*)(buf + 4 * GiB), GiB / 16);
    hipDeviceSynchronize();
    printf("each kernel moved exactly %zu bytes = %zu KiB\n", GiB, GiB / 1024);
    return 0;
}
