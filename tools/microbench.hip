// microbench.hip -- VALU issue rates on gfx950 for the integer ops a 64-bit modular butterfly is built
// from. Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench ; run on the GPU box.
// Each thread runs 8 independent dependency chains per op so latency is hidden; prints cycles per
// wave64-instruction per SIMD (derived with the measured clock from a v_add_u32 loop assumed full rate).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define ITERS 4096
#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[CHAINS], b[CHAINS];
    uint64_t w[CHAINS];
    double f[CHAINS];
    for (int c = 0; c < CHAINS; ++c) {
        a[c] = seed + threadIdx.x * 7 + c; b[c] = seed * 3 + c + 1;
        w[c] = (uint64_t)a[c] * 0x9E3779B97F4A7C15ull; f[c] = 1.0 + a[c] * 1e-9;
    }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(a[c]), "v"(b[c]) : "vcc");
            if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
            if (OP == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
            if (OP == 8) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc"
                                      : "+v"(a[c]), "+v"(b[c]) : "v"(b[(c + 1) % CHAINS]), "v"(a[(c + 1) % CHAINS]) : "vcc");
            if (OP == 9) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
            if (OP == 11) asm volatile("v_cmp_ge_u64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc"
                                       : "+v"(a[c]) : "v"(w[c]), "v"(w[(c + 1) % CHAINS]), "v"(b[c]) : "vcc");
            if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 13) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(f[c]) : "v"(a[c]));
        }
    }
    uint32_t acc = 0;
    for (int c = 0; c < CHAINS; ++c) acc += a[c] + b[c] + (uint32_t)w[c] + (uint32_t)(w[c] >> 32) + (uint32_t)f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
double run(const char* name, uint32_t* d, int blocks, double clk_ghz, int instr_per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = blocks*4 waves/block / (CUs*4 SIMDs) * ITERS*CHAINS*instr
    double waves_per_simd = blocks * 4.0 / (256.0 * 4.0);
    double winstr = waves_per_simd * ITERS * CHAINS * instr_per_iter;
    double ns_per = ms * 1e6 / winstr;
    if (clk_ghz > 0) printf("%-28s %8.3f ms  %6.2f ns/wave-instr/SIMD  = %5.2f cycles @%.2f GHz\n", name, ms, ns_per, ns_per * clk_ghz, clk_ghz);
    return ns_per;
}

int main() {
    uint32_t* d; int blocks = 256 * 8;   // 8 blocks of 256 threads per CU = 8 waves per SIMD
    hipMalloc(&d, blocks * 256 * 4);
    double ns_add = run<0>("v_add_u32", d, blocks, 0, 1);
    ns_add = run<0>("v_add_u32", d, blocks, 0, 1);
    double clk = 2.0 / ns_add;   // assume v_add_u32 issues in 2 cycles per wave64 on SIMD-32
    printf("assuming v_add_u32 = 2 cycles/wave64: effective clock %.2f GHz\n", clk);
    run<0>("v_add_u32", d, blocks, clk, 1);
    run<12>("v_add3_u32", d, blocks, clk, 1);
    run<1>("v_mul_lo_u32", d, blocks, clk, 1);
    run<2>("v_mul_hi_u32", d, blocks, clk, 1);
    run<3>("v_mad_u64_u32", d, blocks, clk, 1);
    run<5>("v_mul_u32_u24", d, blocks, clk, 1);
    run<9>("v_mul_hi_u32_u24", d, blocks, clk, 1);
    run<6>("v_mad_u32_u24", d, blocks, clk, 1);
    run<7>("v_lshl_add_u64", d, blocks, clk, 1);
    run<8>("v_add_co+v_addc_co (pair)", d, blocks, clk, 1);
    run<11>("v_cmp_ge_u64+v_cndmask", d, blocks, clk, 1);
    run<4>("v_fma_f64", d, blocks, clk, 1);
    run<10>("v_mul_f64", d, blocks, clk, 1);
    run<13>("v_cvt_f64_u32", d, blocks, clk, 1);
    return 0;
}
