// valu_rate.hip -- issue rate of the VALU instructions the FP64 modular arithmetic is made of, on gfx950.
// One 1024-thread workgroup per CU (4 waves per SIMD, like the transform kernels), each wave runs ITER x 16 independent
// instances of ONE instruction (16 separate destination registers: no dependency stalls); cycles per wave-instruction
// per SIMD = elapsed shader cycles / (ITER * 16 * 4 waves).   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

#define KERNEL_D(NAME, ASM)                                                                                            \
    __global__ __launch_bounds__(1024) void NAME(double* out, double a, double b, long long* cyc) {                    \
        double r[16];                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) r[i] = a + threadIdx.x + i;                                     \
        __syncthreads();                                                                                               \
        const long long t0 = __builtin_readcyclecounter();                                                             \
        _Pragma("unroll 1") for (int it = 0; it < ITER; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b), "v"(threadIdx.x) : "vcc");            \
        }                                                                                                              \
        const long long t1 = __builtin_readcyclecounter();                                                             \
        double s = 0;                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += r[i];                                                      \
        out[blockIdx.x * 1024 + threadIdx.x] = s;                                                                      \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                               \
    }

KERNEL_D(k_fma, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D(k_mul, "v_mul_f64 %0, %0, %1")
KERNEL_D(k_add, "v_add_f64 %0, %0, %2")
KERNEL_D(k_rndne, "v_rndne_f64 %0, %0")
KERNEL_D(k_floor, "v_floor_f64 %0, %0")
KERNEL_D(k_trunc, "v_trunc_f64 %0, %0")
KERNEL_D(k_fract, "v_fract_f64 %0, %0")
KERNEL_D(k_max, "v_max_f64 %0, %0, %2")
KERNEL_D(k_ldexp, "v_ldexp_f64 %0, %0, 1")
KERNEL_D(k_cvt_u32, "v_cvt_f64_u32 %0, %3")
KERNEL_D(k_cvt_i32, "v_cvt_f64_i32 %0, %3")
KERNEL_D(k_mov64, "v_mov_b64 %0, %1")
KERNEL_D(k_lshl64, "v_lshlrev_b64 %0, 1, %0")
KERNEL_D(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %2")
KERNEL_D(k_madu64, "v_mad_u64_u32 %0, vcc, %3, %3, %0")
KERNEL_D(k_pkfma32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_D(k_pkmul32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_D(k_pkadd32, "v_pk_add_f32 %0, %0, %2")

#define KERNEL_W(NAME, ASM)                                                                                            \
    __global__ __launch_bounds__(1024) void NAME(double* out, double a, double b, long long* cyc) {                    \
        unsigned r[16];                                                                                                \
        const unsigned ua = (unsigned)a, ub = (unsigned)b;                                                             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) r[i] = ua + threadIdx.x + i;                                    \
        __syncthreads();                                                                                               \
        const long long t0 = __builtin_readcyclecounter();                                                             \
        _Pragma("unroll 1") for (int it = 0; it < ITER; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(ua), "v"(ub), "v"(a) : "vcc");  \
        }                                                                                                              \
        const long long t1 = __builtin_readcyclecounter();                                                             \
        unsigned s = 0;                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += r[i];                                                      \
        out[blockIdx.x * 1024 + threadIdx.x] = s;                                                                      \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                               \
    }

KERNEL_W(k_and32, "v_and_b32 %0, %0, %1")
KERNEL_W(k_add32, "v_add_u32 %0, %0, %1")
KERNEL_W(k_addco32, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL_W(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_W(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL_W(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL_W(k_fma32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_W(k_lshl32, "v_lshlrev_b32 %0, 1, %0")
KERNEL_W(k_cvt_to_i32, "v_cvt_i32_f64 %0, %3")
KERNEL_W(k_cvt_f32, "v_cvt_f32_f64 %0, %3")
KERNEL_W(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")

// the lazy forward butterfly as the transforms issue it (8 per thread and stage), with the quotient from v_rndne_f64 and
// from the add-and-subtract-1.5*2^52 form (valid for |quotient| < 2^51 only): 8 instructions each
template <int MAGIC>
__global__ __launch_bounds__(1024) void k_bfly(double* out, double w, double wp, long long* cyc) {
    double r[16];
    const double p = 2251799814045697.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = w + threadIdx.x + i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double x = r[i], y = r[i + 8];
            const double h = y * w;
            const double l = __builtin_fma(y, w, -h);
            double k;
            if (MAGIC) k = __builtin_fma(y, wp, 6755399441055744.0) - 6755399441055744.0;
            else k = __builtin_rint(y * wp);
            const double t = __builtin_fma(-k, p, h) + l;
            r[i] = x + t;
            r[i + 8] = x - t;
        }
        asm volatile("" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(double*, double, double, long long*);
struct Row { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out; long long* cyc;
    CK(hipMalloc(&out, size_t(cus) * 1024 * 8));
    CK(hipMalloc(&cyc, size_t(cus) * 8));
    const Row rows[] = {{"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_rndne_f64", k_rndne}, {"v_floor_f64", k_floor},
                        {"v_trunc_f64", k_trunc}, {"v_fract_f64", k_fract}, {"v_max_f64", k_max}, {"v_ldexp_f64", k_ldexp},
                        {"v_cvt_f64_u32", k_cvt_u32}, {"v_cvt_f64_i32", k_cvt_i32}, {"v_cvt_i32_f64", k_cvt_to_i32}, {"v_cvt_f32_f64", k_cvt_f32},
                        {"v_mov_b64", k_mov64}, {"v_lshlrev_b64", k_lshl64}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_mad_u64_u32", k_madu64},
                        {"v_pk_fma_f32", k_pkfma32}, {"v_pk_mul_f32", k_pkmul32}, {"v_pk_add_f32", k_pkadd32},
                        {"v_and_b32", k_and32}, {"v_add_u32", k_add32}, {"v_add_co_u32", k_addco32}, {"v_cndmask_b32", k_cndmask},
                        {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_fma_f32", k_fma32}, {"v_lshlrev_b32", k_lshl32},
                        {"v_mad_u32_u24", k_mad24}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# %s, %d CUs; one 1024-thread workgroup per CU (4 waves per SIMD), %d x 16 independent instructions per wave\n", prop.name, cus, ITER);
    printf("# %-16s %14s %18s\n", "instruction", "us per launch", "ns per wave-instr per SIMD");
    for (const Row& r : rows) {
        hipLaunchKernelGGL(r.k, dim3(cus), dim3(1024), 0, 0, out, 1.5, 0.25, cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(r.k, dim3(cus), dim3(1024), 0, 0, out, 1.5, 0.25, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / 5;
        printf("  %-16s %14.1f %18.3f\n", r.name, us, us * 1000.0 / (double(ITER) * 16 * 4));
    }
    for (int magic = 0; magic < 2; ++magic) {
        kern_t k = magic ? k_bfly<1> : k_bfly<0>;
        hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 0, 0, out, 1234567.0, 0.3, cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 0, 0, out, 1234567.0, 0.3, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / 5;
        printf("  %-16s %14.1f %18.3f   (per instruction of the 8-instruction butterfly, %s)\n", magic ? "butterfly/magic" : "butterfly/rndne", us,
               us * 1000.0 / (double(ITER / 8) * 8 * 8 * 4), magic ? "quotient = fma(y, w/p, 1.5*2^52) - 1.5*2^52" : "quotient = rndne(y * w/p)");
    }
    printf("# at a 2.4 GHz shader clock a full-rate instruction (4 cycles per wave64) reads 1.67 ns; lower clocks under FP64 load read more\n");
    return 0;
}
