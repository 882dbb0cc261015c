// redeal_shfl_bench.hip -- the wave-private re-deal of the N = 16384 transform (16 doubles per thread: four register-index
// bits change places with four lane-index bits) done two ways:
//   LDS  : the production scheme (ntt_core.hpp redeal_x<PRIVATE>): 16 ds_write_b64 + 16 ds_read_b64 per thread through the
//          wave's own padded LDS block, addresses are immediates, no barrier;
//   SHFL : "innermost stages via wavefront __shfl" (BASELINE north star): a four-step butterfly transposition, per step and
//          register pair  t = lane_bit ? lo : hi;  t = __shfl_xor(t, 1 << lane_bit);  (lane_bit ? lo : hi) = t
//          (2 x v_cndmask + 2 x ds_bpermute_b32 + 2 x v_cndmask per 64-bit pair, 8 pairs per step, 4 steps).
// Each is timed alone and between two blocks of FP64 butterfly-like work (64 fma per value set, like a 4-stage pass),
// 1024 threads per workgroup, one workgroup per CU, 4 waves per SIMD, as in the transform kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ihexl-fpga_amd/csrc tools/experiments/redeal_shfl_bench.hip -o /tmp/redeal_shfl_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "ntt_core.hpp"
using namespace hx;
using G = Geom<14, 4>;

__device__ __forceinline__ void work(double (&v)[16], double w) {      // 4 "stages" of 8 butterflies, 8 FP64 ops each
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int a = (j & ((1 << s) - 1)) | ((j >> s) << (s + 1)), b = a | (1 << s);
            const double h = v[b] * w, l = __builtin_fma(v[b], w, -h), k = __builtin_rint(h * 4.4e-16);
            const double t = __builtin_fma(-k, 2251799814045697.0, h) + l;
            const double x = v[a];
            v[a] = x + t; v[b] = x - t;
        }
}

__device__ __forceinline__ void redeal_lds(double (&v)[16], double* lds, int tid) {
    redeal_x<G, true, false>(v, lds, tid, [](int r, int t) { return G::idxF<6>(r, t); }, [](int r, int t) { return G::idxF<2>(r, t); });
}

// register bit rb <-> lane bit lb, for rb = 0..3 and lb = 2..5 (the same exchange as idxF<6> -> idxF<2> up to a relabelling)
__device__ __forceinline__ void redeal_shfl(double (&v)[16], int lane) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int lb = rb + 2;
        const bool up = (lane >> lb) & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (!(r & (1 << rb))) {
                const int hi = r | (1 << rb);
                double t = up ? v[r] : v[hi];
                t = __shfl_xor(t, 1 << lb, 64);
                if (up) v[r] = t; else v[hi] = t;
            }
    }
}

template <int MODE, bool WORK>
__global__ __launch_bounds__(1024) void k_bench(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    double v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = double(tid * 16 + r);
    for (int it = 0; it < iters; ++it) {
        if (WORK) work(v, 1.0000001);
        if (MODE == 0) redeal_lds(v, lds, tid);
        else if (MODE == 1) redeal_shfl(v, tid & 63);
        asm volatile("" ::: "memory");
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    out[blockIdx.x * 1024 + tid] = s;
}

template <int MODE, bool WORK>
static float run(double* d, int iters) {
    auto k = k_bench<MODE, WORK>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), G::LDS_BYTES, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), G::LDS_BYTES, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;                                          // us per iteration (one workgroup per CU)
}

int main() {
    double* d; hipMalloc(&d, 256 * 1024 * 8);
    const int iters = 2000;
    const float w = run<2, true>(d, iters);
    const float l0 = run<0, false>(d, iters), s0 = run<1, false>(d, iters);
    const float l1 = run<0, true>(d, iters), s1 = run<1, true>(d, iters);
    printf("per iteration, one 1024-thread workgroup per CU (16 waves), microseconds:\n");
    printf("  FP64 work alone (512 ops/thread)                 %.3f\n", w);
    printf("  re-deal alone:            LDS %.3f    __shfl_xor %.3f\n", l0, s0);
    printf("  work + re-deal:           LDS %.3f    __shfl_xor %.3f   (added by the re-deal: %.3f vs %.3f)\n", l1, s1, l1 - w, s1 - w);
    return 0;
}
