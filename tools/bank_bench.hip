// bank_bench.hip -- does the VGPR bank of the source operands change the issue cost of FP64 instructions on gfx950?
// Explicit registers in inline asm; 8 independent accumulators per variant.
//   hipcc --offload-arch=gfx950 -O3 tools/bank_bench.hip -o tools/bank_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
// destination pairs v[20+2i], sources chosen per variant
#define FMA8(A, B, C) \
    "v_fma_f64 v[40:41], v[" A "], v[" B "], v[" C "]\n" "v_fma_f64 v[42:43], v[" A "], v[" B "], v[" C "]\n" \
    "v_fma_f64 v[44:45], v[" A "], v[" B "], v[" C "]\n" "v_fma_f64 v[46:47], v[" A "], v[" B "], v[" C "]\n" \
    "v_fma_f64 v[48:49], v[" A "], v[" B "], v[" C "]\n" "v_fma_f64 v[50:51], v[" A "], v[" B "], v[" C "]\n" \
    "v_fma_f64 v[52:53], v[" A "], v[" B "], v[" C "]\n" "v_fma_f64 v[54:55], v[" A "], v[" B "], v[" C "]\n"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v22","v23"
template <int V>
__global__ __launch_bounds__(256) void k(double* out) {
    asm volatile("v_mov_b32 v8, 0\n v_mov_b32 v9, 0x3ff00000\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n"
                 "v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n" ::: CLOB);
    for (int it = 0; it < ITERS; ++it) {
        if (V == 0) asm volatile(FMA8("8:9", "10:11", "12:13") ::: CLOB);      // pairs start at 8, 10, 12: banks 0,2,0
        if (V == 1) asm volatile(FMA8("8:9", "12:13", "16:17") ::: CLOB);      // all start in bank 0
        if (V == 2) asm volatile(FMA8("8:9", "10:11", "8:9") ::: CLOB);        // repeated operand
        if (V == 3) asm volatile(FMA8("8:9", "12:13", "20:21") ::: CLOB);      // all bank 0, far apart
        if (V == 4) asm volatile(FMA8("8:9", "10:11", "14:15") ::: CLOB);      // bank pairs 01, 23, 23
        if (V == 5) asm volatile(FMA8("10:11", "14:15", "22:23") ::: CLOB);    // all in bank pair 23
    }
    double s; asm volatile("v_add_f64 %0, v[40:41], v[54:55]" : "=v"(s) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V> void run(const char* name, double* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wps = 4, blocks = 256 * wps;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %6.3f ns per v_fma_f64 per SIMD (4 waves/SIMD)\n", name, ms * 1e6 / (double(wps) * ITERS * 8));
}
int main() {
    double* d; hipMalloc(&d, 256 * 4 * 256 * 8);
    run<0>("sources v8, v10, v12 (banks 0,2,0)", d);
    run<1>("sources v8, v12, v16 (all bank 0)", d);
    run<2>("sources v8, v10, v8 (one repeated)", d);
    run<3>("sources v8, v12, v20 (all bank 0)", d);
    run<4>("sources v8, v10, v14 (bank pairs 01,23,23)", d);
    run<5>("sources v10, v14, v22 (all bank pair 23)", d);
    return 0;
}
