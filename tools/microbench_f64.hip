// microbench_f64.hip -- issue cost of the FP64 instruction forms used by f64_arith.hpp on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/microbench_f64.hip -o tools/microbench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
#define CH 8
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed, double s1, double s2) {
    double a[CH], b[CH], c[CH];
    for (int i = 0; i < CH; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; b[i] = seed * 1.0001 + i; c[i] = seed * 0.5 + i * 3; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));                      // 2 distinct VGPR pairs
            if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));          // 3 distinct VGPR pairs
            if (OP == 2) asm volatile("v_fma_f64 %0, -%0, %1, %2" : "+v"(a[i]) : "s"(s1), "v"(c[i]));           // neg, SGPR, VGPR (r = fma(-k,p,h))
            if (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 4) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
            if (OP == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 6) asm volatile("v_add_f64 %0, %0, -%1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 7) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_floor_f64 %0, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_fma_f64 %0, %1, %2, -%0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));         // l = fma(x,w,-h)
            if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(unsigned*)&a[i]) : "v"(*(unsigned*)&b[i]));
            if (OP == 11) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c[i]) : "v"(a[i]), "v"(b[i])); }         // independent dest
        }
    }
    double s = 0; for (int i = 0; i < CH; ++i) s += a[i] + b[i] + c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double* d, int wps) {
    const int blocks = 256 * wps;     // wps blocks of 256 threads per CU = wps waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5, 2.25, 3.5); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5, 2.25, 3.5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD=%d  %6.3f ns per wave-instr per SIMD\n", name, wps, ms * 1e6 / (double(wps) * ITERS * CH));
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int wps : {4, 8}) {
        run<10>("v_add_u32 (reference, full rate)", d, wps);
        run<0>("v_fma_f64 2 VGPR pairs", d, wps);
        run<1>("v_fma_f64 3 VGPR pairs", d, wps);
        run<9>("v_fma_f64 x,w,-h (3 VGPR, neg)", d, wps);
        run<2>("v_fma_f64 -k, SGPR, VGPR", d, wps);
        run<3>("v_mul_f64 VGPR,VGPR", d, wps);
        run<4>("v_mul_f64 SGPR,VGPR", d, wps);
        run<11>("v_mul_f64 independent dest", d, wps);
        run<5>("v_add_f64", d, wps);
        run<6>("v_add_f64 with neg", d, wps);
        run<7>("v_rndne_f64", d, wps);
        run<8>("v_floor_f64", d, wps);
    }
    return 0;
}
