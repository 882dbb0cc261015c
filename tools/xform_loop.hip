// xform_loop.hip -- the round loop of k_ksx_main WITHOUT its memory streams: one 1024-thread workgroup per CU runs R rounds of
// [forward transform (the mod-up configuration of keyswitch_x.hip: lazy period 3, shifted schedule, PRE = 11) + folded
// multiply-accumulate into 2 x 16 accumulators with the "keys" taken from registers] on register-resident data. Only the twiddle
// tables are read from memory. What this loop reaches of the FP64-issue bound is what the transform's own structure (LDS re-deals,
// the cross-wave barrier, twiddle waits) allows; the difference to the real kernel is what its key / input / result streams cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihexl-fpga_amd/csrc -Iinclude [-DHX_FWD_PRIO=1222] [-DXL_MAC=0]
//         tools/xform_loop.hip -o tools/xform_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "hexl_internal.hpp"
#include "ntt_core_f64.hpp"
using namespace hx;
#ifndef XL_MAC
#define XL_MAC 1
#endif
#ifndef XL_PRE
#define XL_PRE 11
#endif
#ifndef XL_LOGE
#define XL_LOGE 4
#endif
using G = Geom<14, XL_LOGE>;
using W = WgNttF64<14, XL_LOGE, 3, XL_PRE, 1>;

__global__ __launch_bounds__(G::T, XL_LOGE == 4 ? 4 : 2) void k_loop(const double* tables, double* out, Mod m, int rounds, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) double ldsx[];
    double v[G::E], acc0[G::E], acc1[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) { v[r] = double((threadIdx.x * 16 + r) * 2654435761u % 1000003u); acc0[r] = 0.0; acc1[r] = 0.0; }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < rounds; ++it) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 toff = (it & 7) * 4 * G::N;
        asm volatile("" : "+s"(toff));
        const double* tb = tables + toff;
        W::template forward<false, false>(v, ldsx, tid, tb, tb + G::N, m);
#pragma unroll
        for (int r = 0; r < G::E; ++r) {
            const double x = v[r];
#if XL_MAC
            const double ka = acc1[(r + 1) % G::E] * 0.25 + 12345.0, kb = acc0[(r + 3) % G::E] * 0.25 - 54321.0;   // |k| <= p/2: "keys" without a load
            acc0[r] = hxf::mac_fold(acc0[r], x, hxf::reduce(ka, m), m);
            acc1[r] = hxf::mac_fold(acc1[r], x, hxf::reduce(kb, m), m);
#endif
            v[r] = hxf::reduce(x, m);          // next round's input (B order read as A order: timing only)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    double s = 0;
#pragma unroll
    for (int r = 0; r < G::E; ++r) s += v[r] + acc0[r] + acc1[r];
    out[blockIdx.x * G::T + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200, grid = argc > 2 ? atoi(argv[2]) : 256;
    const int N = G::N;
    std::vector<double> ht(size_t(8) * 4 * N);
    for (size_t i = 0; i < ht.size(); ++i) ht[i] = double((i * 40503u) % 999983) - 500000;
    double *tb, *out; unsigned long long* cyc;
    hipMalloc(&tb, ht.size() * 8); hipMalloc(&out, size_t(grid) * G::T * 8); hipMalloc(&cyc, grid * 8);
    hipMemcpy(tb, ht.data(), ht.size() * 8, hipMemcpyHostToDevice);
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    hipFuncSetAttribute((const void*)k_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_loop, dim3(grid), dim3(G::T), G::LDS_USED, 0, tb, out, m, rounds, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    std::vector<unsigned long long> hc(grid);
    hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : hc) avg += double(c); avg /= grid;
    // FP64 instructions per round and thread: 14 stages x 8 butterflies x 8 + 4 reductions x 16 x 3 (one dropped: NORED) ... counted
    // from the ISA instead: see the caller; here the nominal 1088 (+ 256 + 48 with the multiply-accumulate stand-in)
    const double instr = 1088.0 + (XL_MAC ? 256.0 + 2 * 48.0 : 0.0) + 48.0;
    const double issue_cycles = instr * (G::T / 64) / 4.0 * 4.0;       // waves per SIMD x 4 cycles
    printf("rounds %d grid %d: %.3f ms, %.0f shader cycles per round (s_memtime clock: 100 MHz units x ...: see ms), %.2f us per round\n",
           rounds, grid, best, avg / rounds, best * 1e3 / rounds);
    printf("issue fraction from the cycle counter: %.3f (nominal issue cycles / shader cycles per round)\n", issue_cycles / (avg / rounds));
    printf("nominal FP64 issue per round: %.0f cycles per SIMD -> at 2.1 GHz %.2f us; issue fraction at 2.1 GHz = %.3f\n", issue_cycles,
           issue_cycles / 2100.0, issue_cycles / 2100.0 / (best * 1e3 / rounds));
    return 0;
}
