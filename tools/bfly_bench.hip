// bfly_bench.hip -- issue cost of the real FP64 butterfly code (f64_arith.hpp) with everything in registers:
// 8 independent lazy butterflies per iteration, no memory traffic. Compare ns per VALU instruction with the
// single-instruction streams of microbench_f64.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihexl-fpga_amd/csrc -Iinclude tools/bfly_bench.hip -o tools/bfly_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "f64_arith.hpp"
using hxf::Mod;
#define ITERS 2000
template <int VARIANT>
__global__ __launch_bounds__(1024) void k(double* out, double seed, Mod m) {
    double v[16], w[8], wp[8];
    for (int i = 0; i < 16; ++i) v[i] = seed * (threadIdx.x + 1) + i * 1000.0;
    for (int i = 0; i < 8; ++i) { w[i] = seed * 3 + i * 77.0 + threadIdx.x; wp[i] = w[i] * m.pinv; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (VARIANT == 0) hxf::ct_bfly_lazy(v[i], v[i + 8], w[i], m);
            if (VARIANT == 1) hxf::ct_bfly(v[i], v[i + 8], w[i], m);
            if (VARIANT == 2) { v[i] = hxf::reduce(v[i], m); v[i + 8] = hxf::reduce(v[i + 8], m); }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
    }
    double s = 0; for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int VARIANT> void run(const char* name, double* d, int wg_per_cu, double valu_per_iter) {
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VARIANT>, dim3(blocks), dim3(1024), 0, 0, d, 1.5, m); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<VARIANT>, dim3(blocks), dim3(1024), 0, 0, d, 1.5, m); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per SIMD per workgroup
    printf("%-30s waves/SIMD=%d  %7.3f ms  %6.3f ns per iteration per wave-slot", name, 4 * wg_per_cu, ms, ms * 1e6 / (double(ITERS) * 4 * wg_per_cu));
    if (valu_per_iter > 0) printf("  = %5.2f ns per VALU instruction", ms * 1e6 / (double(ITERS) * 4 * wg_per_cu * valu_per_iter));
    printf("\n");
}
int main(int argc, char** argv) {
    double* d; hipMalloc(&d, 512 * 1024 * 8);
    // VALU instructions per iteration are read off the ISA (tools/isa_mix.py on the -S output); see DESIGN.md
    const double n0 = argc > 1 ? atof(argv[1]) : 0, n1 = argc > 2 ? atof(argv[2]) : 0, n2 = argc > 3 ? atof(argv[3]) : 0;
    for (int wg = 1; wg <= 2; ++wg) {
        run<0>("8 lazy butterflies", d, wg, n0);
        run<1>("8 strict butterflies", d, wg, n1);
        run<2>("16 reductions", d, wg, n2);
    }
    return 0;
}
