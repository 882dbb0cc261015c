// fp64_power.hip -- what does the chip sustain when NOTHING but FP64 arithmetic runs? A grid that fills every SIMD with 4 waves
// (1024 threads x 256 workgroups, 128 VGPRs like the keyswitch kernels), each thread running MIX of {v_fma_f64, v_mul_f64,
// v_add_f64, v_rndne_f64} on 24 independent register chains -- no memory, no LDS, no barriers -- for `seconds`; prints the
// wave-instruction rate. Run it under tools/power_probe.sh-style rocm-smi sampling: the clock it settles at under the 1400 W
// cap x (its issue fraction = 1) is the power-limited FP64 ceiling that the keyswitch's 0.70 x 2.13 GHz has to be read against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fp64_power tools/fp64_power.hip ; tools/fp64_power [seconds=12] [mix: 0 = FMA only, 1 = butterfly mix]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

template <int MIX>
__global__ __launch_bounds__(1024) void k_burn(double* out, double seed, int iters) {
    double v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = seed + i + threadIdx.x * 1e-3;
    const double a = 1.0000001, b = 0.4999999, p = 2251799814045697.0, pinv = 1.0 / p;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            if (MIX == 0) {
                v[i] = __builtin_fma(v[i], a, b);
            } else {                                               // the butterfly's mix: mul, fma, mul, rndne, fma, add, add, add
                const double h = v[i] * a;
                const double l = __builtin_fma(v[i], a, -h);
                const double k = __builtin_rint(h * pinv);
                const double t = __builtin_fma(-k, p, h) + l;
                v[i] = (v[i] + t) - (b + t);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 12;
    const int mix = argc > 2 ? atoi(argv[2]) : 0;
    double* d = nullptr; (void)hipMalloc((void**)&d, 8);
    const int iters = 20000;
    const double per_launch = 256.0 * 16 * iters * 24 * (mix ? 8 : 1);       // wave-instructions per launch
    auto t0 = std::chrono::steady_clock::now();
    double done = 0;
    std::printf("START\n"); std::fflush(stdout);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int r = 0; r < 4; ++r) {
            if (mix) hipLaunchKernelGGL(k_burn<1>, dim3(256), dim3(1024), 0, 0, d, 1.0, iters);
            else hipLaunchKernelGGL(k_burn<0>, dim3(256), dim3(1024), 0, 0, d, 1.0, iters);
        }
        (void)hipDeviceSynchronize();
        done += 4 * per_launch;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("fp64_power mix=%d: %.3e FP64 wave-instructions/s over %.1f s = %.3f GHz x 1024 SIMDs / 4 cycles\n", mix, done / dt, dt,
                done / dt * 4 / 1024 / 1e9);
    return 0;
}
