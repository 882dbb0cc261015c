// ntt_timeline.hip -- where does a workgroup transform spend its time? Per-wave s_memtime stamps around every phase
// of the production FP64 forward transform (N = 16384, 16 coefficients x 1024 threads), plus the hardware id of
// the CU each workgroup ran on, so the gap between consecutive workgroups on one CU can be measured too.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihexl-fpga_amd/csrc -Iinclude tools/ntt_timeline.hip -o tools/ntt_timeline
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <vector>

#include "ntt_core_f64.hpp"
using namespace hx;

constexpr int NST = 12;
__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
// every coefficient register is pinned at a stamp, so no butterfly can be scheduled across it
#define PIN() _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) asm volatile("" : "+v"(v[r_]))
#define STAMP(i, waits) do { PIN(); asm volatile(waits ::: "memory"); if ((tid & 63) == 0) st[i] = now(); PIN(); } while (0)

template <int LAZY, int STAGGER>
__global__ __launch_bounds__(1024) void k_timeline(double* x, const double* w, const double* wp, Mod m,
                                                   unsigned long long* stamps, unsigned* hwid) {
    using G = Geom<14, 4>;
    using W = WgNttF64<14, 4, LAZY>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    unsigned long long* st = stamps + (size_t(blockIdx.x) * 16 + (tid >> 6)) * NST;
    double v[G::E] = {0};
    // all 256 CUs start their first workgroup together and then stay in lockstep: every round opens with a
    // chip-wide 32 MiB read burst that runs at HBM speed while the FP64 pipes idle. Spreading the first round
    // over one period (STAGGER x 512 cycles) de-phases the CUs for the rest of the launch.
    if (STAGGER && blockIdx.x < 256) {
        const int n = (int(blockIdx.x) * STAGGER) >> 8;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
    }
    STAMP(0, "");
    if (tid == 0) {
        hwid[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        hwid[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
    double* px = x + size_t(blockIdx.x) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(px[G::idxA(r, tid)], m);
    STAMP(1, "s_waitcnt vmcnt(0)");
    fwd_stages_f64<16, 0, 4, 1, 14, LAZY>(v, 0u, w, wp, m);
    STAMP(2, "");
    redeal_x<G, false, false>(v, ldsd, tid, [](int r, int t) { return G::idxF<10>(r, t); }, [](int r, int t) { return G::idxF<6>(r, t); });
    STAMP(3, "s_waitcnt lgkmcnt(0)");
    fwd_stages_f64<16, 0, 4, 5, 14, LAZY>(v, u32(__builtin_amdgcn_readfirstlane(u32(tid) >> 6)), w, wp, m);
    STAMP(4, "");
    redeal_x<G, true, false>(v, ldsd, tid, [](int r, int t) { return G::idxF<6>(r, t); }, [](int r, int t) { return G::idxF<2>(r, t); });
    STAMP(5, "s_waitcnt lgkmcnt(0)");
    fwd_stages_f64<16, 0, 4, 9, 14, LAZY>(v, u32(tid) >> 2, w, wp, m);
    STAMP(6, "");
    redeal_x<G, true, false>(v, ldsd, tid, [](int r, int t) { return G::idxF<2>(r, t); }, [](int r, int t) { return G::idxB(r, t); });
    STAMP(7, "s_waitcnt lgkmcnt(0)");
    W::template fwd_last<0>(v, tid, w, wp, m);
    STAMP(8, "");
#pragma unroll
    for (int r = 0; r < G::E; ++r) px[r * G::T + tid] = v[r];
    STAMP(9, "");
    STAMP(10, "s_waitcnt vmcnt(0)");
}

// persistent variant (the shape of k_ntt_fwd_p, ntt.hip): one workgroup per CU walks the batch, next input requested
// into spare registers right after the conversion of the current one; u64 in / out like the product kernel
__global__ __launch_bounds__(1024) void k_timeline_p(unsigned long long* x, const double* w, const double* wp, Mod m,
                                                     unsigned long long* stamps, unsigned batch) {
    using G = Geom<14, 4>;
    using W = WgNttF64<14, 4, 3>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    unsigned long long raw[G::E];
    {
        const int tid = threadIdx.x;
        const unsigned long long* p0 = x + size_t(blockIdx.x) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (p0 + G::idxA(r, 0))[u32(tid)];
    }
    unsigned it = 0;
#pragma unroll 1
    for (unsigned p = blockIdx.x; p < batch; p += gridDim.x, ++it) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        unsigned long long* st = stamps + ((size_t(blockIdx.x) * 8 + it) * 16 + (tid >> 6)) * NST;
        unsigned long long* px = x + size_t(p) * G::N;
        double v[G::E];
        if ((tid & 63) == 0) st[0] = now();
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(hxf::to_f64(raw[r]), m);
        STAMP(1, "");
        const unsigned pn = p + gridDim.x < batch ? p + gridDim.x : p;
        const unsigned long long* pnx = x + size_t(pn) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (pnx + G::idxA(r, 0))[u32(tid)];
        fwd_stages_f64<16, 0, 4, 1, 14, 3, true>(v, 0u, w, wp, m);
        STAMP(2, "");
        redeal_x<G, false, true>(v, ldsd, tid, [](int r, int t) { return G::idxF<10>(r, t); }, [](int r, int t) { return G::idxF<6>(r, t); });
        STAMP(3, "s_waitcnt lgkmcnt(0)");
        fwd_stages_f64<16, 0, 4, 5, 14, 3, true>(v, u32(__builtin_amdgcn_readfirstlane(u32(tid) >> 6)), w, wp, m);
        STAMP(4, "");
        redeal_x<G, true, false>(v, ldsd, tid, [](int r, int t) { return G::idxF<6>(r, t); }, [](int r, int t) { return G::idxF<2>(r, t); });
        STAMP(5, "s_waitcnt lgkmcnt(0)");
        fwd_stages_f64<16, 0, 4, 9, 14, 3>(v, u32(tid) >> 2, w, wp, m);
        STAMP(6, "");
        redeal_x<G, true, false>(v, ldsd, tid, [](int r, int t) { return G::idxF<2>(r, t); }, [](int r, int t) { return G::idxB(r, t); });
        STAMP(7, "s_waitcnt lgkmcnt(0)");
        W::template fwd_last<0>(v, tid, w, wp, m);
        STAMP(8, "");
        __syncthreads();                                          // stands in for the vote of k_ntt_fwd_p
        STAMP(9, "");
#pragma unroll
        for (int r = 0; r < G::E; ++r) px[G::idxB(r, tid)] = hxf::from_f64(hxf::lift(v[r], m));
        STAMP(10, "");
    }
}

int main(int argc, char** argv) {
    using G = Geom<14, 4>;
    const int N = 16384, batch = argc > 1 ? atoi(argv[1]) : 2048;
    double *d, *w, *wp; unsigned long long* st; unsigned* hw;
    hipMalloc(&d, size_t(batch) * N * 8); hipMalloc(&w, N * 8); hipMalloc(&wp, N * 8);
    hipMalloc(&st, size_t(batch) * 16 * NST * 8); hipMalloc(&hw, batch * 8);
    std::vector<double> h(size_t(batch) * N), hwv(N), hwp(N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = double((i * 2654435761u) % 1000003);
    for (int i = 0; i < N; ++i) { hwv[i] = double((i * 40503u) % 999983) - 500000; hwp[i] = hwv[i] / 2251799814045697.0; }
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(w, hwv.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(wp, hwp.data(), N * 8, hipMemcpyHostToDevice);
    Mod m{2251799814045697.0, 1.0 / 2251799814045697.0};
    for (int variant = 0; variant < 3; ++variant) {
    auto kern = variant == 0 ? k_timeline<3, 0> : variant == 1 ? k_timeline<3, 45> : k_timeline<3, 90>;
    printf("---- first-round stagger: %s\n", variant == 0 ? "none" : variant == 1 ? "half a period" : "one period");
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(batch), dim3(G::T), G::LDS_BYTES, 0, d, w, wp, m, st, hw);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(size_t(batch) * 16 * NST); std::vector<unsigned> id(batch * 2);
    hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(id.data(), hw, id.size() * 4, hipMemcpyDeviceToHost);
    printf("kernel %.3f ms for %d transforms = %.2f us per CU slot (instrumented)\n", ms, batch, ms * 1e3 / (batch / 256.0));
    const char* names[] = {"wait for input", "pass 0 (4 stages)", "cross-wave re-deal (barrier)", "pass 1", "private re-deal", "pass 2", "private re-deal", "pass 3 (2 stages)", "issue stores", "stores retire"};
    // per-phase averages over all waves of all workgroups
    double acc[NST] = {0}, wgspan = 0, skew = 0;
    for (int b = 0; b < batch; ++b) {
        unsigned long long t0 = ~0ull, t1 = 0, e_min = ~0ull;
        for (int wv = 0; wv < 16; ++wv) {
            const unsigned long long* q = &s[(size_t(b) * 16 + wv) * NST];
            for (int i = 0; i < 10; ++i) acc[i] += double(q[i + 1] - q[i]);
            t0 = std::min(t0, q[0]); t1 = std::max(t1, q[10]); e_min = std::min(e_min, q[10]);
        }
        wgspan += double(t1 - t0); skew += double(t1 - e_min);
    }
    double tot = 0;
    for (int i = 0; i < 10; ++i) { acc[i] /= double(batch) * 16; tot += acc[i]; }
    for (int i = 0; i < 10; ++i) printf("  %-32s %8.0f cycles  %5.1f %%\n", names[i], acc[i], 100 * acc[i] / tot);
    printf("  %-32s %8.0f cycles (per wave)\n  workgroup first-start -> last-end %8.0f cycles; last wave ends %0.f cycles after the first\n", "sum", tot, wgspan / batch, skew / batch);
    // gaps between consecutive workgroups of one CU
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
    for (int b = 0; b < batch; ++b) {
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int wv = 0; wv < 16; ++wv) { const unsigned long long* q = &s[(size_t(b) * 16 + wv) * NST]; t0 = std::min(t0, q[0]); t1 = std::max(t1, q[10]); }
        const unsigned hwv2 = id[2 * b];
        const unsigned long long key = ((unsigned long long)(id[2 * b + 1] & 0xf) << 32) | (hwv2 & 0x0000ff00u /*cu, sh*/) | ((hwv2 >> 13) & 0x7) << 16 /*se*/;
        per_cu[key].push_back({t0, t1});
    }
    double gap = 0; size_t ngap = 0;
    for (auto& kv : per_cu) {
        auto& vv = kv.second; std::sort(vv.begin(), vv.end());
        for (size_t i = 1; i < vv.size(); ++i) { gap += double((long long)(vv[i].first - vv[i - 1].second)); ++ngap; }
    }
    printf("  %zu distinct CUs seen; mean gap between a workgroup's last stamp and the next one's first on the same CU: %.0f cycles\n", per_cu.size(), ngap ? gap / ngap : 0.0);
    }
    {   // persistent kernel: 256 workgroups x (batch / 256) polynomials each (<= 8)
        const unsigned pb = batch > 2048 ? 2048 : batch;
        std::vector<unsigned long long> hx(size_t(pb) * N);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (i * 2654435761ull) % 2251799814045697ull;
        unsigned long long* dx; hipMalloc(&dx, hx.size() * 8);
        unsigned long long* stp; hipMalloc(&stp, size_t(256) * 8 * 16 * NST * 8);
        hipFuncSetAttribute((const void*)k_timeline_p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
            hipMemset(stp, 0, size_t(256) * 8 * 16 * NST * 8);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_timeline_p, dim3(256), dim3(G::T), G::LDS_BYTES, 0, dx, w, wp, m, stp, pb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const int iters = pb / 256;
        std::vector<unsigned long long> s(size_t(256) * 8 * 16 * NST);
        hipMemcpy(s.data(), stp, s.size() * 8, hipMemcpyDeviceToHost);
        printf("---- persistent, next input prefetched: %.3f ms for %u transforms = %.2f us per transform per CU (instrumented)\n", ms, pb, ms * 1e3 / iters);
        const char* names[] = {"convert (waits for the prefetched input)", "request next + pass 0", "cross-wave re-deal (barrier)", "pass 1", "private re-deal", "pass 2", "private re-deal", "pass 3 (2 stages)", "end barrier (vote)", "lift + issue stores"};
        for (int it = 0; it < iters; ++it) {
            double acc[NST] = {0}; double tot = 0; double nextgap = 0; int ng = 0;
            for (int b = 0; b < 256; ++b)
                for (int wv = 0; wv < 16; ++wv) {
                    const unsigned long long* q = &s[((size_t(b) * 8 + it) * 16 + wv) * NST];
                    for (int i = 0; i < 10; ++i) acc[i] += double(q[i + 1] - q[i]);
                    if (it + 1 < iters) { const unsigned long long* qn = &s[((size_t(b) * 8 + it + 1) * 16 + wv) * NST]; nextgap += double(qn[0] - q[10]); ++ng; }
                }
            printf("  iteration %d:\n", it);
            for (int i = 0; i < 10; ++i) { acc[i] /= 256.0 * 16; tot += acc[i]; }
            for (int i = 0; i < 10; ++i) printf("    %-44s %8.0f cycles  %5.1f %%\n", names[i], acc[i], 100 * acc[i] / tot);
            printf("    %-44s %8.0f cycles; to the next iteration's first stamp %.0f\n", "sum", tot, ng ? nextgap / ng : 0.0);
        }
    }
    return 0;
}
