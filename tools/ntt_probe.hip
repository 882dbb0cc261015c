// ntt_probe.hip -- ablation of the f64 workgroup NTT: where does a workgroup's time go? Variants drop one
// ingredient at a time (results are then wrong; timing only). Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihexl-fpga_amd/csrc tools/ntt_probe.hip -o tools/ntt_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#ifndef PROBE_MODE_DEFS
#define PROBE_MODE_DEFS
#endif
static __device__ int g_mode_dummy;
#include "ntt_core_f64.hpp"
using namespace hx;

// MODE bits: 1 = no global load, 2 = twiddles from a 1-entry table (always L1/scalar hit), 4 = no LDS re-deal,
//            8 = no global store (one guarded store keeps the values live)
template <int LOGN, int LOGE, int MODE>
struct Probe {
    using G = Geom<LOGN, LOGE>;
    static constexpr int E = G::E;
    template <int PASS>
    __device__ static __forceinline__ void fwd_pass(double (&v)[E], double* lds, int tid, const double* w, const double* wp, const Mod m) {
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = LOGN - (PASS + 1) * LOGE;
            const u32 Gp = (PASS == 0 || (MODE & 2)) ? 0u : (u32(tid) >> LO);
            fwd_stages_f64<E, 0, LOGE, (MODE & 2) ? 1 : PASS * LOGE + 1>(v, Gp, w, wp, m);
            if constexpr (!(MODE & 4)) {
                if constexpr (PASS + 1 < G::P - 1) {
                    constexpr int LO2 = LO - LOGE;
                    redeal_f64<G>(v, lds, tid, [](int r, int t) { return G::template idxF<LO>(r, t); }, [](int r, int t) { return G::template idxF<LO2>(r, t); });
                } else {
                    redeal_f64<G>(v, lds, tid, [](int r, int t) { return G::template idxF<LO>(r, t); }, [](int r, int t) { return G::idxB(r, t); });
                }
            }
            fwd_pass<PASS + 1>(v, lds, tid, w, wp, m);
        } else {
            if constexpr (MODE & 2) {
                fwd_stages_f64<E, 0, G::KL, 1>(v, 0, w, wp, m);
                if constexpr (G::NG > 1) fwd_stages_f64<E, (1 << G::KL), G::KL, 1>(v, 0, w, wp, m);
                if constexpr (G::NG > 2) { fwd_stages_f64<E, 2 * (1 << G::KL), G::KL, 1>(v, 0, w, wp, m); fwd_stages_f64<E, 3 * (1 << G::KL), G::KL, 1>(v, 0, w, wp, m); }
            } else {
                WgNttF64<LOGN, LOGE>::template fwd_last<0>(v, tid, w, wp, m);
            }
        }
    }
};

template <int LOGN, int LOGE, int MODE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_probe(double* x, const double* w, const double* wp, Mod m) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    double* px = x + size_t(blockIdx.x) * G::N;
    double v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = (MODE & 1) ? double(tid * 37 + r) : px[G::idxA(r, tid)];
    Probe<LOGN, LOGE, MODE>::template fwd_pass<0>(v, ldsd, tid, w, wp, m);
    if constexpr (MODE & 8) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < G::E; ++r) s += v[r];
        if (s == 12345.678) px[tid] = s;
    } else {
#pragma unroll
        for (int r = 0; r < G::E; ++r) px[r * G::T + tid] = v[r];
    }
}

// ---- two polynomials per workgroup, interleaved at pass granularity: while one polynomial's LDS re-deal
// (write, barrier, read) is in flight the other one's butterflies keep the FP64 pipe busy.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int LOGN, int LOGE>
struct Dual {
    using G = Geom<LOGN, LOGE>;
    static constexpr int E = G::E;
    template <class F> __device__ static __forceinline__ void wr(double (&v)[E], double* lds, int tid, F f) {
#pragma unroll
        for (int r = 0; r < E; ++r) lds[G::pad(f(r, tid))] = v[r];
    }
    template <class F> __device__ static __forceinline__ void rd(double (&v)[E], double* lds, int tid, F f) {
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = lds[G::pad(f(r, tid))];
    }
    template <int PASS>
    __device__ static __forceinline__ void pass(double (&a)[E], double (&b)[E], double* lds, int tid, const double* w, const double* wp, const Mod m) {
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = LOGN - (PASS + 1) * LOGE;
            const u32 Gp = (PASS == 0) ? 0u : (u32(tid) >> LO);
            auto from = [](int r, int t) { return G::template idxF<LO>(r, t); };
            auto to = [](int r, int t) { if constexpr (PASS + 1 < G::P - 1) return G::template idxF<LO - LOGE>(r, t); else return G::idxB(r, t); };
            fwd_stages_f64<E, 0, LOGE, PASS * LOGE + 1>(a, Gp, w, wp, m);
            lds_barrier();                       // b's reads of the previous exchange have landed everywhere
            wr(a, lds, tid, from);
            lds_barrier();
            rd(a, lds, tid, to);                 // in flight while b computes
            fwd_stages_f64<E, 0, LOGE, PASS * LOGE + 1>(b, Gp, w, wp, m);
            lds_barrier();
            wr(b, lds, tid, from);
            lds_barrier();
            rd(b, lds, tid, to);                 // in flight while a computes its next pass
            pass<PASS + 1>(a, b, lds, tid, w, wp, m);
        } else {
            WgNttF64<LOGN, LOGE>::template fwd_last<0>(a, tid, w, wp, m);
            WgNttF64<LOGN, LOGE>::template fwd_last<0>(b, tid, w, wp, m);
        }
    }
};

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_dual(double* x, const double* w, const double* wp, Mod m) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    double* pa = x + size_t(2 * blockIdx.x) * G::N;
    double* pb = pa + G::N;
    double a[G::E], b[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) a[r] = pa[G::idxA(r, tid)];
#pragma unroll
    for (int r = 0; r < G::E; ++r) b[r] = pb[G::idxA(r, tid)];
    Dual<LOGN, LOGE>::template pass<0>(a, b, ldsd, tid, w, wp, m);
#pragma unroll
    for (int r = 0; r < G::E; ++r) pa[r * G::T + tid] = a[r];
#pragma unroll
    for (int r = 0; r < G::E; ++r) pb[r * G::T + tid] = b[r];
}

// ---- half-size LDS exchange: the re-deal goes through LDS in two halves (index < N/2, then >= N/2), so a
// workgroup needs 70 KiB instead of 140 KiB and TWO workgroups fit on a CU (512 threads x 32 coefficients each,
// 128 VGPRs): one's stalls (loads, barriers, LDS) are covered by the other's FP64 work.
template <int LOGN, int LOGE, int LAZY>
struct HalfX {
    using G = Geom<LOGN, LOGE>;
    static constexpr int E = G::E;
    static constexpr int HALF_WORDS = G::LDS_WORDS / 2;
    template <class FromIdx, class ToIdx>
    __device__ static __forceinline__ void redeal(double (&v)[E], double* lds, int tid, FromIdx from, ToIdx to) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const int idx = from(r, tid);
                if ((idx >> (LOGN - 1)) == half) lds[G::pad(idx & (G::N / 2 - 1))] = v[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const int idx = to(r, tid);
                if ((idx >> (LOGN - 1)) == half) v[r] = lds[G::pad(idx & (G::N / 2 - 1))];
            }
            __syncthreads();
        }
    }
    template <int PASS>
    __device__ static __forceinline__ void pass(double (&v)[E], double* lds, int tid, const double* w, const double* wp, const Mod m) {
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = LOGN - (PASS + 1) * LOGE;
            const u32 Gp = (PASS == 0) ? 0u : (u32(tid) >> LO);
            fwd_stages_f64<E, 0, LOGE, PASS * LOGE + 1, LOGN, LAZY>(v, Gp, w, wp, m);
            if constexpr (PASS + 1 < G::P - 1) {
                constexpr int LO2 = LO - LOGE;
                redeal(v, lds, tid, [](int r, int t) { return G::template idxF<LO>(r, t); }, [](int r, int t) { return G::template idxF<LO2>(r, t); });
            } else {
                redeal(v, lds, tid, [](int r, int t) { return G::template idxF<LO>(r, t); }, [](int r, int t) { return G::idxB(r, t); });
            }
            pass<PASS + 1>(v, lds, tid, w, wp, m);
        } else {
            WgNttF64<LOGN, LOGE, LAZY>::template fwd_last<0>(v, tid, w, wp, m);
        }
    }
};

template <int LOGN, int LOGE, int LAZY, int WPS, int STAGGER = 0>
__global__ __launch_bounds__(1 << (LOGN - LOGE), WPS) void k_halfx(double* x, const double* w, const double* wp, Mod m) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    double* px = x + size_t(blockIdx.x) * G::N;
    double v[G::E];
    // two co-resident workgroups that start together run their phases in lockstep and never overlap; a one-time
    // stagger of the first wave of workgroups (every second one waits ~half a transform) breaks the symmetry
    if (blockIdx.x < 512 && (blockIdx.x & 1)) {
        for (int i = 0; i < STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = px[G::idxA(r, tid)];
    HalfX<LOGN, LOGE, LAZY>::template pass<0>(v, ldsd, tid, w, wp, m);
#pragma unroll
    for (int r = 0; r < G::E; ++r) px[r * G::T + tid] = v[r];
}

template <int LOGN, int LOGE, int LAZY, int WPS, int STAGGER = 0>
float run_halfx(double* d, const double* w, const double* wp, int batch, const char* name) {
    using G = Geom<LOGN, LOGE>;
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    const int lb = (int)(HalfX<LOGN, LOGE, LAZY>::HALF_WORDS * 8 + 64);
    auto kern = k_halfx<LOGN, LOGE, LAZY, WPS, STAGGER>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lb);
    int nblk = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, kern, G::T, lb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(batch), dim3(G::T), lb, 0, d, w, wp, m);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(batch), dim3(G::T), lb, 0, d, w, wp, m);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("E=%2d HALFX %-28s blocks/CU=%d  %8.3f ms  %6.2f us/NTT/CU  %6.2f M NTT/s\n", 1 << LOGE, name, nblk, ms, ms * 1e3 / (batch / 256.0), batch / ms / 1e3);
    return ms;
}

// ---- persistent workgroups with register prefetch: one workgroup per CU walks a contiguous range of
// polynomials; the next polynomial's coefficients are requested before the last (short) register pass of the
// current one, so the HBM latency and the workgroup relaunch gap disappear from the critical path.
template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_persist(double* x, const double* w, const double* wp, Mod m, int total) {
    using G = Geom<LOGN, LOGE>;
    using W = WgNttF64<LOGN, LOGE, LAZY>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int first = blockIdx.x * per, last = min(total, first + per);
    if (first >= last) return;
    double v[G::E], nxt[G::E];
    {
        const double* px = x + size_t(first) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = px[G::idxA(r, tid)];
    }
    for (int item = first; item < last; ++item) {
        const double* tw = w + opaque_zero();
        const double* twp = wp + opaque_zero();
        // all passes but the last
        W::template fwd_pass_until_last<0>(v, ldsd, tid, tw, twp, m);
        if (item + 1 < last) {
            const double* pn = x + size_t(item + 1) * G::N;
#pragma unroll
            for (int r = 0; r < G::E; ++r) nxt[r] = pn[G::idxA(r, tid)];
        }
        W::template fwd_last<0>(v, tid, tw, twp, m);
        double* px = x + size_t(item) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) px[r * G::T + tid] = v[r];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = nxt[r];
    }
}

template <int LOGN, int LOGE, int LAZY>
float run_persist(double* d, const double* w, const double* wp, int batch, int grid) {
    using G = Geom<LOGN, LOGE>;
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    const int lb = (int)G::LDS_BYTES;
    auto kern = k_persist<LOGN, LOGE, LAZY>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::T), lb, 0, d, w, wp, m, batch);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(G::T), lb, 0, d, w, wp, m, batch);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("E=%2d PERSIST lazy=%d grid=%d                               %8.3f ms  %6.2f us/NTT/CU  %6.2f M NTT/s\n", 1 << LOGE, (int)LAZY, grid, ms, ms * 1e3 / (batch / 256.0), batch / ms / 1e3);
    return ms;
}

template <int LOGN, int LOGE>
float run_dual(double* d, const double* w, const double* wp, int batch) {
    using G = Geom<LOGN, LOGE>;
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    hipFuncSetAttribute((const void*)k_dual<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_dual<LOGN, LOGE>), dim3(batch / 2), dim3(G::T), G::LDS_BYTES, 0, d, w, wp, m);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_dual<LOGN, LOGE>), dim3(batch / 2), dim3(G::T), G::LDS_BYTES, 0, d, w, wp, m);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("E=%2d DUAL (two polynomials per workgroup)                 %8.3f ms  %6.2f us/NTT/CU    %6.2f M NTT/s\n", 1 << LOGE, ms, ms * 1e3 / (batch / 256.0), batch / ms / 1e3);
    return ms;
}

template <int LOGN, int LOGE, int MODE>
float run(double* d, const double* w, const double* wp, int batch, const char* name) {
    using G = Geom<LOGN, LOGE>;
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    hipFuncSetAttribute((const void*)k_probe<LOGN, LOGE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<LOGN, LOGE, MODE>), dim3(batch), dim3(G::T), G::LDS_BYTES, 0, d, w, wp, m);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_probe<LOGN, LOGE, MODE>), dim3(batch), dim3(G::T), G::LDS_BYTES, 0, d, w, wp, m);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("E=%2d mode %2d %-44s %8.3f ms  %6.2f us/WG-slot  %6.2f M NTT/s\n", 1 << LOGE, MODE, name, ms, ms * 1e3 / (batch / 256.0), batch / ms / 1e3);
    return ms;
}

// two streams, each running the half-LDS kernel on half of the batch: workgroups of the two grids share CUs but
// start at unrelated times, i.e. co-resident workgroups are NOT in lockstep. Compare with one stream.
template <int LOGN, int LOGE, int LAZY>
void run_two_streams(double* d, const double* w, const double* wp, int batch) {
    using G = Geom<LOGN, LOGE>;
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    const int lb = (int)(HalfX<LOGN, LOGE, LAZY>::HALF_WORDS * 8 + 64);
    auto kern = k_halfx<LOGN, LOGE, LAZY, 4, 0>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lb);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, eb; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&eb);
    for (int variant = 0; variant < 2; ++variant) {
        hipDeviceSynchronize();
        hipEventRecord(e0, sa);
        hipStreamWaitEvent(sb, e0, 0);
        for (int i = 0; i < 5; ++i) {
            if (variant == 0) {
                hipLaunchKernelGGL(kern, dim3(batch), dim3(G::T), lb, sa, d, w, wp, m);
            } else {
                hipLaunchKernelGGL(kern, dim3(batch / 2), dim3(G::T), lb, sa, d, w, wp, m);
                hipLaunchKernelGGL(kern, dim3(batch / 2), dim3(G::T), lb, sb, d + size_t(batch / 2) * G::N, w, wp, m);
            }
        }
        hipEventRecord(eb, sb);
        hipStreamWaitEvent(sa, eb, 0);
        hipEventRecord(e1, sa);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("E=%2d HALFX %s: %8.3f ms  %6.2f M NTT/s\n", 1 << LOGE, variant ? "two streams (desynchronised co-residents)" : "one stream", ms, batch / ms / 1e3);
    }
}

int main() {
    const int N = 16384, batch = 2048;
    double *d, *w, *wp;
    hipMalloc(&d, size_t(batch) * N * 8); hipMalloc(&w, N * 8); hipMalloc(&wp, N * 8);
    std::vector<double> h(size_t(batch) * N), hw(N), hwp(N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = double((i * 2654435761u) % 1000003);
    for (int i = 0; i < N; ++i) { hw[i] = double((i * 40503u) % 999983) - 500000; hwp[i] = hw[i] / 2251799814045697.0; }
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(wp, hwp.data(), N * 8, hipMemcpyHostToDevice);
    run<14, 4, 0>(d, w, wp, batch, "full");
    run<14, 4, 1>(d, w, wp, batch, "no global load");
    run<14, 4, 8>(d, w, wp, batch, "no global store");
    run<14, 4, 9>(d, w, wp, batch, "no global load/store");
    run<14, 4, 2>(d, w, wp, batch, "twiddles from one cache line");
    run<14, 4, 4>(d, w, wp, batch, "no LDS re-deal");
    run<14, 4, 11>(d, w, wp, batch, "no load/store, cheap twiddles");
    run<14, 4, 15>(d, w, wp, batch, "ALU only (no load/store/twiddle/LDS)");
    // correctness of the dual kernel vs the single one
    {
        std::vector<double> r1(h.size()), r2(h.size());
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        { const int lb = (int)Geom<14, 4>::LDS_BYTES; hipFuncSetAttribute((const void*)k_probe<14, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lb); }
        Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
        const size_t LB = Geom<14, 4>::LDS_BYTES;
        hipLaunchKernelGGL((k_probe<14, 4, 0>), dim3(batch), dim3(1024), LB, 0, d, w, wp, m);
        hipMemcpy(r1.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        { const int lb = (int)Geom<14, 4>::LDS_BYTES; hipFuncSetAttribute((const void*)k_dual<14, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lb); }
        hipLaunchKernelGGL((k_dual<14, 4>), dim3(batch / 2), dim3(1024), LB, 0, d, w, wp, m);
        hipMemcpy(r2.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < r1.size(); ++i) bad += r1[i] != r2[i];
        printf("dual vs single mismatches: %zu\n", bad);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    run_dual<14, 4>(d, w, wp, batch);
    {   // lazy single-shot reference for the persistent variants
        using G4 = Geom<14, 4>;
        Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
        std::vector<double> r1(size_t(8) * N), r2(size_t(8) * N);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        const size_t LB = G4::LDS_BYTES;
        hipLaunchKernelGGL((k_probe<14, 4, 0>), dim3(8), dim3(1024), LB, 0, d, w, wp, m);
        hipMemcpy(r1.data(), d, r1.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        auto kp = k_persist<14, 4, 0>;
        { const int lb = (int)LB; hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, lb); }
        hipLaunchKernelGGL(kp, dim3(3), dim3(1024), LB, 0, d, w, wp, m, 8);
        hipMemcpy(r2.data(), d, r2.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < r1.size(); ++i) bad += r1[i] != r2[i];
        printf("persist vs single mismatches: %zu\n", bad);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    run_persist<14, 4, 0>(d, w, wp, batch, 256);
    run_persist<14, 4, 3>(d, w, wp, batch, 256);
    run_persist<14, 4, 3>(d, w, wp, batch, 512);
    run_halfx<14, 4, 3, 1>(d, w, wp, batch, "E=16 lazy half-exchange, 1 WG");
    run_halfx<14, 4, 3, 8>(d, w, wp, batch, "E=16 lazy half-exchange, 64 VGPR cap (2 WG/CU)");
    run_halfx<14, 4, 3, 8, 3>(d, w, wp, batch, "E=16 lazy half-exch, 64 VGPR, stagger 3");
    {   // correctness of the half-exchange kernel vs the single one (E=32 outputs are in the E=32 B order: compare sorted sums)
        std::vector<double> r1(size_t(4) * N), r2(size_t(4) * N);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
        const size_t LB5 = Geom<14, 5>::LDS_BYTES;
        { const int lb = (int)LB5; hipFuncSetAttribute((const void*)k_probe<14, 5, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lb); }
        hipLaunchKernelGGL((k_probe<14, 5, 0>), dim3(4), dim3(512), LB5, 0, d, w, wp, m);
        hipMemcpy(r1.data(), d, r1.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        const int lbh = (int)(HalfX<14, 5, 0>::HALF_WORDS * 8 + 64);
        auto kh = k_halfx<14, 5, 0, 4>;
        hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, lbh);
        hipLaunchKernelGGL(kh, dim3(4), dim3(512), lbh, 0, d, w, wp, m);
        hipMemcpy(r2.data(), d, r2.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < r1.size(); ++i) bad += r1[i] != r2[i];
        printf("halfx vs single (E=32) mismatches: %zu\n", bad);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    run_halfx<14, 5, 0, 4>(d, w, wp, batch, "strict, 128 VGPR cap");
    run_halfx<14, 5, 3, 4>(d, w, wp, batch, "lazy, 128 VGPR cap");
    run_halfx<14, 5, 3, 2>(d, w, wp, batch, "lazy, 256 VGPR (1 WG/CU)");
    run_halfx<14, 5, 0, 4, 1>(d, w, wp, batch, "strict, stagger 1");
    run_halfx<14, 5, 0, 4, 2>(d, w, wp, batch, "strict, stagger 2");
    run_halfx<14, 5, 0, 4, 3>(d, w, wp, batch, "strict, stagger 3");
    run_halfx<14, 5, 0, 4, 5>(d, w, wp, batch, "strict, stagger 5");
    run_halfx<14, 5, 3, 4, 3>(d, w, wp, batch, "lazy, stagger 3");
    run_two_streams<14, 5, 3>(d, w, wp, batch);
    run_two_streams<14, 5, 0>(d, w, wp, batch);
    run<14, 5, 0>(d, w, wp, batch, "full");
    run<14, 5, 9>(d, w, wp, batch, "no global load/store");
    run<14, 5, 15>(d, w, wp, batch, "ALU only");
    return 0;
}
