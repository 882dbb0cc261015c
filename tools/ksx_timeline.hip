// ksx_timeline.hip -- where do the slot-major keyswitch kernels (hexl-fpga_amd/csrc/keyswitch_x.hip) spend their time?
// Runs k_ksx_special / k_ksx_main on synthetic in-range data with the KX_TIMELINE stamps compiled in and prints, per
// round type, the average per-wave cycles between phase boundaries, the workgroup span and the kernel time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihexl-fpga_amd/csrc -Iinclude tools/ksx_timeline.hip -o tools/ksx_timeline
//   tools/ksx_timeline [nb=256] [L=7]
#define KX_TIMELINE 1
#include "keyswitch_x.hip"

#include <algorithm>
#include <vector>

static double rnd(unsigned long long& s, double p) {   // centred pseudo-random residue
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return double((long long)((s >> 11) % (unsigned long long)p)) - p / 2;
}

int main(int argc, char** argv) {
#ifndef KX_LOGE
#define KX_LOGE 4      // the production geometry (16 coefficients x 1024 threads); 5 = 32 x 512
#endif
    using G = Geom<14, KX_LOGE>;
    const u32 nb = argc > 1 ? atoi(argv[1]) : 256, L = argc > 2 ? atoi(argv[2]) : 7, K = L + 1, N = G::N;
    const double p0 = 2251799814045697.0;
    std::vector<KsModF64> mods(K);
    for (u32 i = 0; i < K; ++i) {
        KsModF64& f = mods[i];
        f.m.p = p0 + 32768.0 * 2 * i; f.m.pinv = 1.0 / f.m.p;
        f.sc.n = 12345.0; f.sc.n_p = f.sc.n / f.m.p; f.sc.nw = -54321.0; f.sc.nw_p = f.sc.nw / f.m.p;
        f.msf = 777777.0; f.msf_p = f.msf / f.m.p; f.fix = 1000.0; f.half = 1125899907022848.0;
    }
    unsigned long long seed = 42;
    std::vector<double> tables(size_t(K) * 4 * N), keys(size_t(L) * (L + 1) * 2 * N);
    for (u32 i = 0; i < K; ++i)
        for (int blk = 0; blk < 4; blk += 2)
            for (u32 r = 0; r < N; ++r) {
                const double w = rnd(seed, mods[i].m.p);
                tables[(size_t(i) * 4 + blk) * N + r] = w;
                tables[(size_t(i) * 4 + blk + 1) * N + r] = w / mods[i].m.p;
            }
    for (auto& k : keys) k = rnd(seed, p0);
    std::vector<u64> t(size_t(nb) * L * N), res(size_t(nb) * 2 * L * N);
    for (auto& x : t) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; x = (seed >> 13) % (u64)p0; }
    for (auto& x : res) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; x = (seed >> 13) % (u64)p0; }

    u32* dflag = nullptr;
    hipMalloc((void**)&dflag, 4); hipMemset(dflag, 0, 4);
    KsArgsX a;
    KsModF64* dm; double *dt, *dk, *dc, *ds; u64 *dtt, *dres; unsigned long long* dst;
    hipMalloc(&dm, K * sizeof(KsModF64)); hipMalloc(&dt, tables.size() * 8); hipMalloc(&dk, keys.size() * 8);
    hipMalloc(&dc, size_t(nb) * L * N * 8); hipMalloc(&ds, size_t(nb) * 2 * N * 8);
    hipMalloc(&dtt, t.size() * 8); hipMalloc(&dres, res.size() * 8);
    const size_t nst = size_t(nb) * L * (G::T / 64) * KX_NST;
    hipMalloc(&dst, nst * 8);
    hipMemcpy(dm, mods.data(), K * sizeof(KsModF64), hipMemcpyHostToDevice);
    hipMemcpy(dt, tables.data(), tables.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dk, keys.data(), keys.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtt, t.data(), t.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dres, res.data(), res.size() * 8, hipMemcpyHostToDevice);
    a.mods = dm; a.tables = dt; a.keys = dk; a.c = dc; a.s = ds; a.t_target = dtt; a.result = dres;
    a.L = L; a.K = K; a.nb = nb; a.stamps = dst; a.key_stride = 2u << 14; a.alias = 0; a.range_flag = dflag;
    a.nsel = L; a.selmap = 0xFEDCBA9876543210ull;                    // every limb in one launch (plans of one tier)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kin = k_ksx_intt<14, KX_LOGE, 3>;
    hipFuncSetAttribute((const void*)kin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kin, dim3((nb * L + 7) / 8 * 8), dim3(G::T), G::LDS_USED, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("==== k_ksx_intt: grid %u, %.1f us = %.1f us per round of 256 workgroups\n", nb * L, ms * 1e3, ms * 1e3 / ((nb * L + 255) / 256));
    }
    // the kernels bench.py's workload runs: lazy period 3, SKIP (moduli of one size: no range reduction of c_d / s')
    auto ksp = k_ksx_special<14, KX_LOGE, 3, true>;
    auto kmn = k_ksx_main<14, KX_LOGE, 3, false, true>;
    hipFuncSetAttribute((const void*)ksp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED);
    hipFuncSetAttribute((const void*)kmn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED);
    const int W = G::T / 64;
    std::vector<unsigned long long> s(nst);
    for (int which = 0; which < 2; ++which) {
        const u32 grid = ((which ? nb * L : nb) + 7) / 8 * 8;   // one item per workgroup: stamps stay per item
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(dst, 0, nst * 8);
            hipEventRecord(e0);
            if (which) hipLaunchKernelGGL(kmn, dim3(grid), dim3(G::T), G::LDS_USED, 0, a);
            else       hipLaunchKernelGGL(ksp, dim3(grid), dim3(G::T), G::LDS_USED, 0, a);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(s.data(), dst, nst * 8, hipMemcpyDeviceToHost);
        printf("==== %s: grid %u, %.1f us (instrumented) = %.1f us per round of 256 workgroups\n", which ? "k_ksx_main" : "k_ksx_special",
               grid, ms * 1e3, ms * 1e3 / ((grid + 255) / 256));
        const int rounds = L + 2, endi = 4 * rounds;
        // per-round phase averages
        std::vector<double> ph(size_t(rounds) * 4, 0.0); std::vector<double> cnt(size_t(rounds) * 4, 0.0);
        double span = 0, skew = 0, diag_ld = 0, diag_mac = 0; size_t nd = 0;
        for (u32 g = 0; g < grid; ++g) {
            unsigned long long t0 = ~0ull, t1 = 0, emin = ~0ull;
            for (int w = 0; w < W; ++w) {
                const unsigned long long* q = &s[(size_t(g) * W + w) * KX_NST];
                // collect the non-zero stamps in order
                std::vector<std::pair<int, unsigned long long>> ev;
                for (int i = 0; i <= endi; ++i) if (q[i]) ev.push_back({i, q[i]});
                for (size_t e = 0; e + 1 < ev.size(); ++e) {
                    const int idx = ev[e].first;
                    ph[idx] += double(ev[e + 1].second - ev[e].second); cnt[idx] += 1;
                }
                if (ev.empty()) continue;
                unsigned long long start = ev.front().second;
                if (which && q[60]) { start = q[60]; diag_ld += double(q[61] - q[60]); diag_mac += double(ev.front().second - q[61]); ++nd; }
                t0 = std::min(t0, start); t1 = std::max(t1, ev.back().second); emin = std::min(emin, ev.back().second);
            }
            span += double(t1 - t0); skew += double(t1 - emin);
        }
        if (argc > 3) {   // per-wave Gantt of ONE workgroup (argv[3]): stamps relative to the workgroup's first, in cycles
            const u32 g = (u32)atoi(argv[3]) % grid;
            unsigned long long base = ~0ull;
            for (int w = 0; w < W; ++w) for (int i = 0; i < KX_NST; ++i) { const unsigned long long x = s[(size_t(g) * W + w) * KX_NST + i]; if (x && x < base) base = x; }
            printf("  gantt of item %u, first stamp at absolute cycle %llu (columns: stamp index; rows: wave, SIMD = wave %% 4)\n        ", g, base);
            std::vector<int> cols;
            if (which) { cols.push_back(60); cols.push_back(61); }
            for (int i = 0; i <= endi; ++i) { bool any = false; for (int w = 0; w < W; ++w) any |= s[(size_t(g) * W + w) * KX_NST + i] != 0; if (any) cols.push_back(i); }
            for (int c : cols) printf("%8d", c);
            printf("\n");
            for (int w = 0; w < W; ++w) {
                printf("  w%02d s%d ", w, w % 4);
                for (int c : cols) printf("%8lld", (long long)(s[(size_t(g) * W + w) * KX_NST + c] - base));
                printf("\n");
            }
        }
        if (which) printf("  diagonal: load t_i -> B %8.0f cycles, multiply-accumulate (+ first input request) %8.0f cycles\n", diag_ld / nd, diag_mac / nd);
        const char* nm_sp[4] = {"wait for input + reduce | inverse + store s' (last two)", "forward transform", "multiply-accumulate", ""};
        const char* nm_mn[4] = {"wait for input + reduce", "forward transform", "multiply-accumulate (up) | result epilogue (down)", ""};
        for (int r = 0; r < rounds; ++r)
            for (int k = 0; k < 4; ++k)
                if (cnt[r * 4 + k] > 0)
                    printf("  round %d  %-64s %8.0f cycles (%.0f waves)\n", r, which ? nm_mn[k] : nm_sp[k], ph[r * 4 + k] / cnt[r * 4 + k], cnt[r * 4 + k]);
        printf("  workgroup first stamp -> last stamp %8.0f cycles; last wave ends %.0f cycles after the first\n", span / grid, skew / grid);
    }
    return 0;
}
