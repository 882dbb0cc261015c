// geom_bench.hip -- forward FP64 transform throughput of the candidate N=16384 geometries (c -> u, as k_ksf_ntt_up).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihexl-fpga_amd/csrc -Iinclude tools/geom_bench.hip -o tools/geom_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "ntt_core_f64.hpp"
using namespace hx;
template <int LOGE, bool HALFX, int MINW, int STAG = 0>
__global__ __launch_bounds__(1 << (14 - LOGE), MINW) void k_fwd(const double* c, double* u, const double* tb, Mod m) {
    // first-wave stagger: co-resident workgroups that start together stay in lockstep
    if (STAG == 1 && blockIdx.x < 512 && (blockIdx.x & 1)) for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(8);
    if (STAG == 2 && blockIdx.x >= 256 && blockIdx.x < 512) for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(8);
    if (STAG == 3 && blockIdx.x < 512 && ((blockIdx.x >> 3) & 1)) for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(8);
    if (STAG == 4) {   // co-resident workgroups sit in different wave slots: give one of them the higher issue priority
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));   // HW_ID.wave_id
        if (hw & 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    }
    if (STAG == 5) { if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    using G = Geom<14, LOGE>;
    using W = WgNttF64<14, LOGE, true, HALFX>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const double* cd = c + size_t(blockIdx.x) * G::N;
    double v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(cd[G::idxA(r, tid)], m);
    W::template forward<true>(v, ldsd, tid, tb, tb + G::N, m);
    double* dst = u + size_t(blockIdx.x) * G::N;
    int ts = tid;
    asm volatile("" : "+v"(ts));                 // keeps the 32 store addresses from being computed (and spilled) up front
#pragma unroll
    for (int r = 0; r < G::E; ++r) dst[r * G::T + ts] = v[r];
}
template <int LOGE, bool HALFX, int MINW, int STAG = 0>
void run(const char* name, const double* c, double* u, const double* tb, int batch) {
    using G = Geom<14, LOGE>;
    const int lds = HALFX ? (int)G::LDS_HALF_BYTES : (int)G::LDS_BYTES;
    auto kern = k_fwd<LOGE, HALFX, MINW, STAG>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int nblk = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, kern, G::T, lds);
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(batch), dim3(G::T), lds, 0, c, u, tb, m); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(batch), dim3(G::T), lds, 0, c, u, tb, m);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-46s WG/CU=%d  %7.3f ms  %6.2f us per transform per CU  %6.2f M/s\n", name, nblk, ms, ms * 1e3 / (batch / 256.0), batch / ms / 1e3);
}
int main() {
    const int N = 16384, batch = 4096;
    double *c, *u, *tb;
    hipMalloc(&c, size_t(batch) * N * 8); hipMalloc(&u, size_t(batch) * N * 8); hipMalloc(&tb, 2 * N * 8);
    std::vector<double> h(size_t(batch) * N), ht(2 * N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = double((i * 2654435761u) % 1000003);
    for (int i = 0; i < N; ++i) { ht[i] = double((i * 40503u) % 999983) - 500000; ht[N + i] = ht[i] / 2251799814045697.0; }
    hipMemcpy(c, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(tb, ht.data(), ht.size() * 8, hipMemcpyHostToDevice);
    run<4, false, 1>("16 x 1024, full exchange, 1 WG/CU", c, u, tb, batch);
    run<4, true, 8>("16 x 1024, half exchange, 64 VGPR, 2 WG/CU", c, u, tb, batch);
    run<4, true, 8, 4>("  same, priority by wave slot", c, u, tb, batch);
    run<4, true, 8, 5>("  same, priority by workgroup parity", c, u, tb, batch);
    run<4, true, 1>("16 x 1024, half exchange, 128 VGPR", c, u, tb, batch);
    run<5, false, 1>("32 x 512, full exchange (256 VGPR)", c, u, tb, batch);
    return 0;
}
