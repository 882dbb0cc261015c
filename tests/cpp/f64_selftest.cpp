// f64_selftest.cpp -- host-side validation of hexl-fpga_amd/csrc/f64_arith.hpp (the exact FP64 modular
// arithmetic the keyswitch kernels run on gfx950). IEEE-754 double mul/add/fma/rint are correctly rounded
// on both x86 (-mfma) and gfx950, so agreement with exact __int128 arithmetic here proves the bounds the
// device code relies on. Checks primitives on adversarial operands, then whole transforms against the
// oracle's canonical NTT / INTT.  Build: tests/cpp/Makefile (g++ -O2 -mfma -ffp-contract=off).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hexl-fpga_amd/csrc/f64_arith.hpp"
#include "../../oracle/hexl_oracle.h"

typedef __int128 i128;
static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { if (++failures < 20) { std::printf("FAIL line %d: ", __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static int64_t centred(i128 v, int64_t p) { int64_t r = (int64_t)(v % p); if (r < 0) r += p; return r; }   // canonical actually

static void test_prime(uint64_t p) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    const int64_t P = (int64_t)p;
    std::vector<int64_t> edge = {0, 1, -1, P / 2, -(P / 2), P / 2 + 1, -(P / 2) - 1, P - 1, -(P - 1), P, -P, P + 1,
                                 (3 * P) / 2, -(3 * P) / 2, P / 2 + 2, P / 3, 2 * P / 3};
    auto pick = [&](int64_t bound) -> int64_t {
        uint64_t r = rnd();
        if ((r & 7) == 0) { int64_t e = edge[(r >> 3) % edge.size()]; if (e > bound) e = bound; if (e < -bound) e = -bound; return e; }
        return (int64_t)(rnd() % (2 * (uint64_t)bound + 1)) - bound;
    };
    for (int it = 0; it < 400000; ++it) {
        // reduce: any |x| < 2^53
        int64_t x = pick(((int64_t)1 << 53) - 1);
        double r = hxf::reduce((double)x, m);
        CHECK(r == (double)(int64_t)r && (int64_t)r >= -(P / 2) - 2 && (int64_t)r <= P / 2 + 2 && centred((i128)x - (int64_t)r, P) == 0,
              "reduce p=%lu x=%ld r=%.0f", p, x, r);
        // lift: centred -> canonical
        int64_t c = pick(P / 2 + 2);
        double l = hxf::lift((double)c, m);
        CHECK(l == (double)centred(c, P), "lift p=%lu c=%ld got %.0f", p, c, l);
        // mul_shoup: |x| <= 1.5p, |w| <= p/2
        int64_t xs = pick((3 * P) / 2), w = pick(P / 2);
        double wp = (double)w / (double)p;
        double t = hxf::mul_shoup((double)xs, (double)w, wp, m);
        double bound = (0.5 + (double)(xs < 0 ? -xs : xs) / (2.0 * p)) * p + 1;
        CHECK(t == (double)(int64_t)t && centred((i128)xs * w - (int64_t)t, P) == 0 && (t <= bound && t >= -bound),
              "mul_shoup p=%lu x=%ld w=%ld t=%.0f", p, xs, w, t);
        // mul_mod: |a|,|b| <= p/2 + 2
        int64_t a = pick(P / 2 + 2), b = pick(P / 2 + 2);
        double u = hxf::mul_mod((double)a, (double)b, m);
        CHECK(u == (double)(int64_t)u && centred((i128)a * b - (int64_t)u, P) == 0 && u <= 0.7 * p + 2 && u >= -0.7 * p - 2,
              "mul_mod p=%lu a=%ld b=%ld u=%.0f", p, a, b, u);
        // conversions
        uint64_t v = rnd() % p;
        CHECK(hxf::from_f64(hxf::to_f64(v)) == v, "convert %lu", v);
        CHECK(hxf::to_f64_lt52(v) == hxf::to_f64(v), "convert (OR form) %lu", v);
    }
}

// full transforms with the butterflies of f64_arith.hpp vs the oracle's canonical transforms
static void test_transforms(uint64_t n, uint64_t p) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t *inv0 = blk.data(), *roots = blk.data() + 2 * n;
    auto centre = [&](uint64_t v) { return hxf::reduce(hxf::to_f64(v), m); };
    std::vector<uint64_t> x(n), ref;
    orc_fill_splitmix(x.data(), n, p ^ n, p);
    x[0] = p - 1; x[1] = 0; x[2] = p / 2; x[3] = p / 2 + 1;
    // forward
    ref = x; orc_ks_ntt(ref.data(), n, p, roots);
    std::vector<double> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = centre(x[i]);
    for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1)
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(roots[mm + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) hxf::ct_bfly(v[j], v[j + t], w, m);
        }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64(hxf::lift(v[i], m)) == ref[i], "fwd n=%lu p=%lu i=%lu", n, p, i);
    // inverse (table from index 0, then * n^-1)
    ref = x; orc_ks_intt(ref.data(), n, p, inv0);
    for (uint64_t i = 0; i < n; ++i) v[i] = centre(x[i]);
    uint64_t acc = 0;
    for (uint64_t mm = n >> 1, t = 1; mm >= 1; mm >>= 1, t <<= 1) {
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(inv0[acc + i]), wp = w / (double)p;
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) hxf::gs_bfly(v[j], v[j + t], w, wp, m);
        }
        acc += mm;
    }
    const double ninv = centre(orc_invmod(n, p)), ninv_p = ninv / (double)p;
    for (uint64_t i = 0; i < n; ++i) {
        double r = hxf::reduce(hxf::mul_shoup(v[i], ninv, ninv_p, m), m);
        CHECK(hxf::from_f64(hxf::lift(r, m)) == ref[i], "inv n=%lu p=%lu i=%lu", n, p, i);
    }
}

// LAZY regime: replay the device schedules (ntt_core_f64.hpp) on the host, tracking the largest magnitude any
// value reaches (must stay < 2^53) and checking results against the oracle.
// chains of the folded multiply-accumulate (mac_fold) against exact integers: x up to the un-reduced transform bound of the
// tier, keys up to p/2, both random and pinned at the extremes with the signs that push the accumulator outwards
static double g_fold_max = 0;
static double g_fold_strict_max = 0, g_fold_strict_inner = 0;
// the STRICT tier of mac_fold (keyswitch_x.hip, strict kernels): moduli up to 2^52, x a reduced transform output, the first accumulator
// the reduced d == i term; tracks the accumulator and the two intermediates whose exactness the fma / the addition rely on
static void test_mac_fold_strict(uint64_t p) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    const int64_t P = (int64_t)p;
    const int64_t xmax = P / 2 + 2, kmax = P / 2;
    for (int chain = 0; chain < 40000; ++chain) {
        int64_t a0 = (int64_t)(rnd() % (uint64_t)(P + 5)) - (P / 2 + 2);
        if (chain % 8 == 1) a0 = P / 2 + 2; else if (chain % 8 == 2) a0 = -(P / 2 + 2);
        double acc = (double)a0;
        i128 exact = a0;
        const int mode = chain % 4;
        for (int term = 0; term < 16; ++term) {
            int64_t x, k;
            if (mode == 0) { x = (int64_t)(rnd() % (2 * (uint64_t)xmax + 1)) - xmax; k = (int64_t)(rnd() % (2 * (uint64_t)kmax + 1)) - kmax; }
            else {
                x = xmax - (int64_t)(rnd() % 1024); k = kmax - (int64_t)(rnd() % 1024);
                const bool up = mode == 1 ? true : mode == 2 ? (acc >= 0) : (term & 1);
                if (!up) x = -x;
                if (rnd() & 1) { x = -x; k = -k; }
            }
            {   // the intermediates of mac_fold, replayed
                const double h = (double)x * (double)k, l = __builtin_fma((double)x, (double)k, -h);
                const double K = __builtin_rint(__builtin_fma(acc, m.pinv, h * m.pinv));
                const double inner = __builtin_fma(-K, m.p, h), sum = acc + l;
                const double ai = inner < 0 ? -inner : inner, as = sum < 0 ? -sum : sum;
                if (ai > g_fold_strict_inner) g_fold_strict_inner = ai;
                if (as > g_fold_strict_inner) g_fold_strict_inner = as;
                // (h - K p) + (acc + l) must be the exact value of x k + acc - K p: both parts are integers below 2^53
                CHECK(inner == (double)(int64_t)inner && sum == (double)(int64_t)sum &&
                      (i128)(int64_t)inner + (i128)(int64_t)sum == (i128)x * k + (i128)(int64_t)acc - (i128)(int64_t)K * P,
                      "strict mac_fold intermediates p=%lu x=%ld k=%ld", p, x, k);
            }
            acc = hxf::mac_fold(acc, (double)x, (double)k, m);
            exact += (i128)x * k;
            const double a = acc < 0 ? -acc : acc;
            if (a / (double)p > g_fold_strict_max) g_fold_strict_max = a / (double)p;
            CHECK(acc == (double)(int64_t)acc && centred(exact - (int64_t)acc, P) == 0 && a <= 0.9 * (double)p,
                  "strict mac_fold p=%lu chain %d term %d acc=%.0f", p, chain, term, acc);
        }
        const double r = hxf::reduce(acc, m);
        CHECK((int64_t)r >= -(P / 2) - 2 && (int64_t)r <= P / 2 + 2 && centred(exact - (int64_t)r, P) == 0, "strict mac_fold final reduce p=%lu", p);
    }
}
static void test_mac_fold(uint64_t p, double c) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    const int64_t P = (int64_t)p;
    const int64_t xmax = (int64_t)(c * (double)p), kmax = P / 2;
    for (int chain = 0; chain < 20000; ++chain) {
        double acc = 0;
        i128 exact = 0;
        const int mode = chain % 4;
        for (int term = 0; term < 16; ++term) {
            int64_t x, k;
            if (mode == 0) { x = (int64_t)(rnd() % (2 * (uint64_t)xmax + 1)) - xmax; k = (int64_t)(rnd() % (2 * (uint64_t)kmax + 1)) - kmax; }
            else {
                x = xmax - (int64_t)(rnd() % 1024); k = kmax - (int64_t)(rnd() % 1024);
                const bool up = mode == 1 ? true : mode == 2 ? (acc >= 0) : (term & 1);
                if (!up) x = -x;
                if (rnd() & 1) { x = -x; k = -k; }
            }
            acc = hxf::mac_fold(acc, (double)x, (double)k, m);
            exact += (i128)x * k;
            const double a = acc < 0 ? -acc : acc;
            if (a / (double)p > g_fold_max) g_fold_max = a / (double)p;
            CHECK(acc == (double)(int64_t)acc && centred(exact - (int64_t)acc, P) == 0 && a <= 1.6 * (double)p,
                  "mac_fold p=%lu chain %d term %d acc=%.0f", p, chain, term, acc);
        }
    }
}

static double g_max_abs = 0;
static inline void track(double x) { double a = x < 0 ? -x : x; if (a > g_max_abs) g_max_abs = a; }

static void test_transforms_lazy(uint64_t n, uint64_t p, bool adversarial, int period = 3) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    int logn = 0; while ((1ull << logn) < n) ++logn;
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t *inv0 = blk.data(), *roots = blk.data() + 2 * n;
    auto centre = [&](uint64_t v) { return hxf::reduce(hxf::to_f64(v), m); };
    std::vector<uint64_t> x(n), ref;
    orc_fill_splitmix(x.data(), n, p ^ (n + 7), p);
    if (adversarial) for (uint64_t i = 0; i < n; ++i) x[i] = (i & 1) ? p / 2 : p / 2 + 1;   // extreme centred magnitudes
    ref = x; orc_ks_ntt(ref.data(), n, p, roots);
    std::vector<double> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = centre(x[i]);
    int s = 1;
    for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1, ++s) {
        const bool red = hxf::lazy_fwd_reduce_after(s, logn, period);
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(roots[mm + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                if (red) hxf::ct_bfly(v[j], v[j + t], w, m); else hxf::ct_bfly_lazy(v[j], v[j + t], w, m);
                track(v[j]); track(v[j + t]);
            }
        }
    }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64(hxf::lift(v[i], m)) == ref[i], "lazy fwd n=%lu p=%lu i=%lu", n, p, i);
    {   // the same schedule WITHOUT the reduction after the last stage (mod-up / mod-down kernels), followed by its
        // two consumers: mul_mod with a centred key (k_ksf_mac) and mul_shoup of (prod - w) (k_ksf_moddown)
        std::vector<double> u(n);
        for (uint64_t i = 0; i < n; ++i) u[i] = centre(x[i]);
        int s2 = 1;
        for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1, ++s2) {
            const bool red = hxf::lazy_fwd_reduce_after(s2, 0, period);
            for (uint64_t i = 0; i < mm; ++i) {
                const double w = centre(roots[mm + i]);
                for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                    if (red) hxf::ct_bfly(u[j], u[j + t], w, m); else hxf::ct_bfly_lazy(u[j], u[j + t], w, m);
                    track(u[j]); track(u[j + t]);
                }
            }
        }
        std::vector<uint64_t> key(n), pr(n);
        orc_fill_splitmix(key.data(), n, p ^ 0x5151, p);
        orc_fill_splitmix(pr.data(), n, p ^ 0x7373, p);
        const uint64_t msf = adversarial ? p / 2 + 1 : key[0];
        const double msf_c = centre(msf), msf_p = msf_c / (double)p;
        for (uint64_t i = 0; i < n; ++i) {
            CHECK(hxf::from_f64(hxf::lift(hxf::reduce(u[i], m), m)) == ref[i], "unreduced tail n=%lu p=%lu i=%lu", n, p, i);
            const uint64_t kv = adversarial ? ((i & 1) ? p / 2 : p / 2 + 1) : key[i];
            const double prod = hxf::mul_mod(u[i], centre(kv), m);
            track(prod);
            CHECK(hxf::from_f64(hxf::lift(hxf::reduce(prod, m), m)) == orc_mulmod(ref[i], kv, p), "mac on unreduced u n=%lu p=%lu i=%lu", n, p, i);
            const double in = centre(pr[i]) - u[i];
            const double out = hxf::mul_shoup(in, msf_c, msf_p, m);
            track(in); track(out);
            const uint64_t diff = (pr[i] + p - ref[i]) % p;
            CHECK(hxf::from_f64(hxf::lift(hxf::reduce(out, m), m)) == orc_mulmod(diff, msf, p), "moddown on unreduced w n=%lu p=%lu i=%lu", n, p, i);
        }
    }
    // inverse, last stage fused with the scaling exactly as inv_stages_f64 does
    ref = x; orc_ks_intt(ref.data(), n, p, inv0);
    for (uint64_t i = 0; i < n; ++i) v[i] = centre(x[i]);
    const double ninv = centre(orc_invmod(n, p)), ninv_p = ninv / (double)p;
    const double nw = centre(orc_mulmod(orc_invmod(n, p), inv0[n - 2], p)), nw_p = nw / (double)p;
    uint64_t acc = 0;
    for (uint64_t mm = n >> 1, t = 1; mm >= 1; mm >>= 1, t <<= 1) {
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(inv0[acc + i]), wp = w / (double)p;
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                if (mm > 1) {
                    hxf::gs_bfly_lazy(v[j], v[j + t], w, wp, m);
                } else {
                    const double sum = v[j] + v[j + t], dif = v[j] - v[j + t];
                    track(sum); track(dif);
                    v[j] = hxf::reduce(hxf::mul_shoup(sum, ninv, ninv_p, m), m);
                    v[j + t] = hxf::reduce(hxf::mul_shoup(dif, nw, nw_p, m), m);
                }
                track(v[j]); track(v[j + t]);
            }
        }
        acc += mm;
    }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64(hxf::lift(v[i], m)) == ref[i], "lazy inv n=%lu p=%lu i=%lu", n, p, i);
}

// ---- round 4 ---------------------------------------------------------------------------------------------------
// (1) forward transform on the SHIFTED schedule with inputs that are not centred: canonical residues of a neighbouring modulus,
//     0 <= x < rho p (keyswitch_x.hip SKIP kernels), without the reduction after the last stage, followed by mac_fold;
// (2) the mod-down epilogue with un-reduced accumulators: mul_shoup(acc - w) at |acc| = 1.7p, |w| up to the tail bound;
// (3) inverse transforms without the w/p table (gs_bfly_lazy_nowp + the strict stage), canonical inputs taken as they are.
static void test_round4(uint64_t n, uint64_t p, int period, double rho, bool adversarial) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    int logn = 0; while ((1ull << logn) < n) ++logn;
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t* roots = blk.data() + 2 * n;
    auto centre = [&](uint64_t v) { return hxf::reduce(hxf::to_f64(v), m); };
    const uint64_t top = (uint64_t)(rho * (double)p) - 1;              // inputs in [0, rho p)
    std::vector<uint64_t> x(n), ref(n);
    for (uint64_t i = 0; i < n; ++i) x[i] = adversarial ? top - (i & 3) : rnd() % (top + 1);
    for (uint64_t i = 0; i < n; ++i) ref[i] = x[i] % p;
    orc_ks_ntt(ref.data(), n, p, roots);
    std::vector<double> u(n);
    for (uint64_t i = 0; i < n; ++i) u[i] = hxf::to_f64(x[i]);            // as they are
    int s = 1;
    for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1, ++s) {
        // the device drops the periodic reduction that falls on the last stage (ntt_core_f64.hpp NORED): up to `period` un-reduced stages
        const bool red = hxf::lazy_fwd_reduce_after(s, 0, period, 1) && s != logn;
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(roots[mm + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                if (red) hxf::ct_bfly(u[j], u[j + t], w, m); else hxf::ct_bfly_lazy(u[j], u[j + t], w, m);
                track(u[j]); track(u[j + t]);
            }
        }
    }
    std::vector<uint64_t> key(n);
    orc_fill_splitmix(key.data(), n, p ^ 0x4444, p);
    for (uint64_t i = 0; i < n; ++i) {
        CHECK(hxf::from_f64(hxf::lift(hxf::reduce(u[i], m), m)) == ref[i], "shifted fwd n=%lu p=%lu i=%lu", n, p, i);
        // mac_fold on the un-reduced output, accumulator at its bound with the sign that pushes outwards
        const uint64_t kv = adversarial ? ((i & 1) ? p / 2 : p / 2 + 1) : key[i];
        const double kc = centre(kv);
        const double acc0 = ((u[i] < 0) != (kc < 0) ? -1.0 : 1.0) * (double)(uint64_t)(1.6 * (double)p);
        const double acc = hxf::mac_fold(acc0, u[i], kc, m);
        track(acc);
        const i128 exact = (i128)(int64_t)acc0 + (i128)(int64_t)u[i] * (int64_t)kc;
        CHECK(acc == (double)(int64_t)acc && centred(exact - (int64_t)acc, (int64_t)p) == 0 && (acc < 0 ? -acc : acc) <= 1.7 * (double)p,
              "mac_fold on shifted tail n=%lu p=%lu i=%lu acc=%.0f", n, p, i, acc);
        // mod-down epilogue: (acc - w) * msf with |acc| = 1.7p un-reduced and w = this un-reduced value (>= the real tail bound)
        if (period == 3 || (u[i] < 0 ? -u[i] : u[i]) <= 2.2 * (double)p) {
            const double a17 = (u[i] < 0 ? 1.0 : -1.0) * (double)(uint64_t)(1.7 * (double)p);
            const double wv = (u[i] < 0 ? -u[i] : u[i]) <= 2.2 * (double)p ? u[i] : (u[i] < 0 ? -1.0 : 1.0) * (double)(uint64_t)(2.2 * (double)p);
            const uint64_t msf = adversarial ? p / 2 + 1 : key[(i + 1) % n];
            const double msf_c = centre(msf), in = a17 - wv;
            const double out = hxf::mul_shoup(in, msf_c, msf_c / (double)p, m);
            track(in); track(out);
            CHECK(out == (double)(int64_t)out && centred((i128)(int64_t)in * (int64_t)msf_c - (int64_t)out, (int64_t)p) == 0,
                  "moddown on un-reduced acc n=%lu p=%lu i=%lu", n, p, i);
        }
    }
    // centred s' (|y| <= rho p / 2) straight into the standard schedule
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t y = adversarial ? ((i & 1) ? 1 : -1) * (int64_t)(top / 2) : (int64_t)(rnd() % (top + 1)) - (int64_t)(top / 2);
        u[i] = (double)y;
        ref[i] = (uint64_t)centred((i128)y, (int64_t)p);
    }
    orc_ks_ntt(ref.data(), n, p, roots);
    s = 1;
    for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1, ++s) {
        const bool red = hxf::lazy_fwd_reduce_after(s, 0, period, 0);
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(roots[mm + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                if (red) hxf::ct_bfly(u[j], u[j + t], w, m); else hxf::ct_bfly_lazy(u[j], u[j + t], w, m);
                track(u[j]); track(u[j + t]);
            }
        }
    }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64(hxf::lift(hxf::reduce(u[i], m), m)) == ref[i], "centred s' fwd n=%lu p=%lu i=%lu", n, p, i);
}

// ---- round 6: the X schedules (f64_arith.hpp XSCHED_TABLE; ntt_core_f64.hpp fwd_stages_f64<..., XS>) ------------------------------------
// A forward transform that range-reduces only the added operand, in front of the stages the table names; inputs as the kernels feed them
// (shift 1: canonical residues of a neighbouring modulus, [0, rho p); shift 0: centred values up to 0.5 rho p), no reduction after the last
// stage, followed by the consumer the schedule was chosen for: mac_fold at |acc| = 1.6p (down = 0) or the mod-down epilogue
// mul_shoup(acc - w) at |acc| = 1.7p (down = 1), signs chosen to push outwards.
static double g_xs_max = 0, g_xs_tail = 0;
static void test_xsched(uint64_t n, uint64_t p, int period, int shift, bool down, double rho, bool adversarial) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    int logn = 0; while ((1ull << logn) < n) ++logn;
    const unsigned mask = hxf::xsched_mask(period, shift, down, logn);
    CHECK(mask != 0u, "no X schedule for period %d shift %d down %d stages %d", period, shift, (int)down, logn);
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t* roots = blk.data() + 2 * n;
    auto centre = [&](uint64_t v) { return hxf::reduce(hxf::to_f64(v), m); };
    auto trackx = [&](double x) { const double a = x < 0 ? -x : x; if (a > g_xs_max) g_xs_max = a; };
    const uint64_t top = (uint64_t)(rho * (double)p) - 1;
    std::vector<uint64_t> ref(n);
    std::vector<double> u(n);
    for (uint64_t i = 0; i < n; ++i) {
        int64_t y;
        if (shift) y = (int64_t)(adversarial ? top - (i & 3) : rnd() % (top + 1));
        else y = adversarial ? ((i & 1) ? 1 : -1) * (int64_t)(top / 2) : (int64_t)(rnd() % (top + 1)) - (int64_t)(top / 2);
        u[i] = (double)y;
        ref[i] = (uint64_t)centred((i128)y, (int64_t)p);
        if ((int64_t)ref[i] < 0) ref[i] += p;
    }
    orc_ks_ntt(ref.data(), n, p, roots);
    int s = 1;
    for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1, ++s) {
        const int op = hxf::xsched_op(mask, s);
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(roots[mm + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                if (op >= 1) u[j] = hxf::reduce(u[j], m);
                if (op == 2) u[j + t] = hxf::reduce(u[j + t], m);
                hxf::ct_bfly_lazy(u[j], u[j + t], w, m);
                trackx(u[j]); trackx(u[j + t]);
            }
        }
    }
    std::vector<uint64_t> key(n);
    orc_fill_splitmix(key.data(), n, p ^ 0x6666, p);
    for (uint64_t i = 0; i < n; ++i) {
        const double au = u[i] < 0 ? -u[i] : u[i];
        if (au / (double)p > g_xs_tail) g_xs_tail = au / (double)p;
        CHECK(u[i] == (double)(int64_t)u[i] && hxf::from_f64(hxf::lift(hxf::reduce(u[i], m), m)) == ref[i], "X schedule fwd n=%lu p=%lu (period %d shift %d down %d) i=%lu",
              n, p, period, shift, (int)down, i);
        if (!down) {
            const uint64_t kv = adversarial ? ((i & 1) ? p / 2 : p / 2 + 1) : key[i];
            const double kc = centre(kv);
            const double acc0 = ((u[i] < 0) != (kc < 0) ? -1.0 : 1.0) * (double)(uint64_t)(1.6 * (double)p);
            const double acc = hxf::mac_fold(acc0, u[i], kc, m);
            trackx(acc);
            const i128 exact = (i128)(int64_t)acc0 + (i128)(int64_t)u[i] * (int64_t)kc;
            CHECK(acc == (double)(int64_t)acc && centred(exact - (int64_t)acc, (int64_t)p) == 0 && (acc < 0 ? -acc : acc) <= 1.7 * (double)p,
                  "mac_fold on an X-schedule tail n=%lu p=%lu i=%lu acc=%.0f", n, p, i, acc);
        } else {
            const double a17 = (u[i] < 0 ? 1.0 : -1.0) * (double)(uint64_t)(1.7 * (double)p);
            const uint64_t msf = adversarial ? p / 2 + 1 : key[(i + 1) % n];
            const double msf_c = centre(msf), in = a17 - u[i];
            const double out = hxf::mul_shoup(in, msf_c, msf_c / (double)p, m);
            trackx(in); trackx(out);
            CHECK(out == (double)(int64_t)out && centred((i128)(int64_t)in * (int64_t)msf_c - (int64_t)out, (int64_t)p) == 0,
                  "mod-down epilogue on an X-schedule tail n=%lu p=%lu i=%lu", n, p, i);
        }
    }
}

// inverse without the w/p table, canonical input words as they are; lazy = the lazy kernels' schedule, else the strict one
static void test_inverse_nowp(uint64_t n, uint64_t p, bool lazy, bool adversarial, double rho = 1.0) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t* inv0 = blk.data();
    auto centre = [&](uint64_t v) { return hxf::reduce(hxf::to_f64(v), m); };
    std::vector<uint64_t> x(n), ref;
    orc_fill_splitmix(x.data(), n, p ^ (n + 99), p);
    if (adversarial) for (uint64_t i = 0; i < n; ++i) x[i] = (i & 1) ? p - 1 : ((i & 2) ? 0 : p - 2);
    if (rho > 1.0) {                                                   // words of the standalone inverse's fast path: below rho p, as they are
        const uint64_t top = (uint64_t)(rho * (double)p) - 1;
        for (uint64_t i = 0; i < n; ++i) x[i] = adversarial ? ((i & 1) ? top : ((i & 2) ? 0 : top - 1)) : rnd() % (top + 1);
    }
    ref = x;
    for (auto& v : ref) v %= p;
    orc_ks_intt(ref.data(), n, p, inv0);
    std::vector<double> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = hxf::to_f64_lt52(x[i]);
    const double ninv = centre(orc_invmod(n, p)), ninv_p = ninv / (double)p;
    const double nw = centre(orc_mulmod(orc_invmod(n, p), inv0[n - 2], p)), nw_p = nw / (double)p;
    uint64_t acc = 0;
    int gs = 1;
    for (uint64_t mm = n >> 1, t = 1; mm >= 1; mm >>= 1, t <<= 1, ++gs) {
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(inv0[acc + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                if (mm > 1) {
                    const double dd = v[j] - v[j + t];
                    track(dd); track(v[j] + v[j + t]);
                    if (lazy && gs != hxf::INV_NOWP_STRICT_STAGE) hxf::gs_bfly_lazy_nowp(v[j], v[j + t], w, m);
                    else hxf::gs_bfly_nowp(v[j], v[j + t], w, m);
                } else {
                    const double sum = v[j] + v[j + t], dif = v[j] - v[j + t];
                    track(sum); track(dif);
                    v[j] = hxf::reduce(hxf::mul_shoup(sum, ninv, ninv_p, m), m);
                    v[j + t] = hxf::reduce(hxf::mul_shoup(dif, nw, nw_p, m), m);
                }
                track(v[j]); track(v[j + t]);
            }
        }
        acc += mm;
    }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64(hxf::lift(v[i], m)) == ref[i], "nowp inverse n=%lu p=%lu lazy=%d i=%lu", n, p, (int)lazy, i);
}

// round 6: the I schedules (f64_arith.hpp ISCHED_TABLE; ntt_core_f64.hpp inv_bfly<..., IS>): the table-free inverse with the range
// reductions the table names, decided per stage and per history of the butterfly's inputs (both sum outputs / both product outputs of
// the previous stage; at the stages that open a register pass of the (logn, loge) geometry the table holds one decision for both),
// canonical words below rho p as they are, the fused last stage as the device runs it
static double g_is_max = 0;
static void test_isched(uint64_t n, int loge, uint64_t p, int period, bool adversarial, double rho) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    int logn = 0; while ((1ull << logn) < n) ++logn;
    const unsigned long long mask = hxf::isched_mask(period, logn, loge);
    CHECK(mask != 0ull, "no I schedule for period %d logn %d loge %d", period, logn, loge);
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t* inv0 = blk.data();
    auto centre = [&](uint64_t v) { return hxf::reduce(hxf::to_f64(v), m); };
    auto tracki = [&](double x) { const double a = x < 0 ? -x : x; if (a > g_is_max) g_is_max = a; };
    const uint64_t top = (uint64_t)(rho * (double)p) - 1;
    std::vector<uint64_t> x(n), ref;
    for (uint64_t i = 0; i < n; ++i) x[i] = adversarial ? ((i & 1) ? top : ((i & 2) ? 0 : top - 1)) : rnd() % (top + 1);
    ref = x;
    for (auto& v : ref) v %= p;
    orc_ks_intt(ref.data(), n, p, inv0);
    std::vector<double> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = hxf::to_f64_lt52(x[i]);
    const double ninv = centre(orc_invmod(n, p)), ninv_p = ninv / (double)p;
    const double nw = centre(orc_mulmod(orc_invmod(n, p), inv0[n - 2], p)), nw_p = nw / (double)p;
    uint64_t acc = 0;
    int gs = 1;
    for (uint64_t mm = n >> 1, t = 1; mm >= 1; mm >>= 1, t <<= 1, ++gs) {
        const bool opens = hxf::isched_opens_pass(logn, loge, gs);
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(inv0[acc + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                const double sum = v[j] + v[j + t], dif = v[j] - v[j + t];
                tracki(sum); tracki(dif);
                if (mm > 1) {
                    // history: the previous stage paired (k, k + t / 2): its product outputs sit where bit t / 2 of the index is set
                    const bool from_products = !opens && t > 1 && (j & (t >> 1)) != 0;
                    const int bits = hxf::isched_bits(mask, gs) >> (from_products ? 2 : 0);
                    const double pr = hxf::mul_mod(dif, w, m);
                    tracki(pr);
                    v[j] = (bits & 1) ? hxf::reduce(sum, m) : sum;
                    v[j + t] = (bits & 2) ? hxf::reduce(pr, m) : pr;
                } else {
                    v[j] = hxf::reduce(hxf::mul_shoup(sum, ninv, ninv_p, m), m);
                    v[j + t] = hxf::reduce(hxf::mul_shoup(dif, nw, nw_p, m), m);
                }
            }
        }
        acc += mm;
    }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64(hxf::lift(v[i], m)) == ref[i], "I schedule inverse n=%lu loge=%d p=%lu (period %d) i=%lu", n, loge, p, period, i);
}

// STRICT butterflies at moduli in [2^52, STRICT_NTT_MAX_Q) -- the standalone _NTT / _INTT fast path of ntt.hip for SURVEY 8d's
// q = 2^52 + 393217: forward (ct_bfly) and table-free inverse (gs_bfly_nowp, fused last stage on mul_shoup) with the device's own
// conversions (reduce(to_f64(raw)) in, from_f64_53(lift()) out), against the oracle, tracking every intermediate magnitude
static double g_wide_max = 0;
static inline void trackw(double x) { double a = x < 0 ? -x : x; if (a > g_wide_max) g_wide_max = a; }
static void test_strict_wide(uint64_t n, uint64_t p, bool adversarial) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t *inv0 = blk.data(), *roots = blk.data() + 2 * n;
    auto centre = [&](uint64_t v) { return v > p / 2 ? (double)v - (double)p : (double)v; };      // k_ntt_prepare
    std::vector<uint64_t> x(n), ref;
    orc_fill_splitmix(x.data(), n, p ^ (n + 7), p);
    if (adversarial) for (uint64_t i = 0; i < n; ++i) x[i] = (i & 1) ? p - 1 : ((i & 2) ? p / 2 + 1 : p / 2);
    // word <-> double conversions at the top of the range
    for (uint64_t v : {p - 1, p - 2, (uint64_t)1 << 52, ((uint64_t)1 << 52) + 1, ((uint64_t)1 << 52) - 1, (uint64_t)0, (uint64_t)1, p / 2})
        CHECK(hxf::from_f64_53(hxf::to_f64(v)) == v, "from_f64_53 %lu", v);
    for (int i = 0; i < 100000; ++i) { const uint64_t v = rnd() >> 11; CHECK(hxf::from_f64_53(hxf::to_f64(v)) == v, "from_f64_53 %lu", v); }
    // forward
    ref = x; orc_ks_ntt(ref.data(), n, p, roots);
    std::vector<double> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = hxf::reduce(hxf::to_f64(x[i]), m);
    for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1)
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(roots[mm + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                const double tt = hxf::mul_mod(v[j + t], w, m);
                trackw(tt); trackw(v[j] + tt); trackw(v[j] - tt);
                hxf::ct_bfly(v[j], v[j + t], w, m);
                trackw(v[j]); trackw(v[j + t]);
            }
        }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64_53(hxf::lift(v[i], m)) == ref[i], "wide fwd n=%lu p=%lu i=%lu", n, p, i);
    // inverse, table-free, fused last stage
    ref = x; orc_ks_intt(ref.data(), n, p, inv0);
    for (uint64_t i = 0; i < n; ++i) v[i] = hxf::reduce(hxf::to_f64(x[i]), m);
    const double ninv = centre(orc_invmod(n, p)), ninv_p = ninv / (double)p;
    const double nw = centre(orc_mulmod(orc_invmod(n, p), inv0[n - 2], p)), nw_p = nw / (double)p;
    uint64_t acc = 0;
    for (uint64_t mm = n >> 1, t = 1; mm >= 1; mm >>= 1, t <<= 1) {
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(inv0[acc + i]);
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                const double sum = v[j] + v[j + t], dif = v[j] - v[j + t];
                trackw(sum); trackw(dif);
                if (mm > 1) {
                    trackw(hxf::mul_mod(dif, w, m));
                    hxf::gs_bfly_nowp(v[j], v[j + t], w, m);
                } else {
                    const double a = hxf::mul_shoup(sum, ninv, ninv_p, m), b = hxf::mul_shoup(dif, nw, nw_p, m);
                    trackw(a); trackw(b);
                    v[j] = hxf::reduce(a, m);
                    v[j + t] = hxf::reduce(b, m);
                }
                trackw(v[j]); trackw(v[j + t]);
            }
        }
        acc += mm;
    }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64_53(hxf::lift(v[i], m)) == ref[i], "wide inv n=%lu p=%lu i=%lu", n, p, i);
}
// the SEMI-STRICT forward schedule (f64_arith.hpp ct_bfly_semi, ntt_core_f64.hpp fwd_stages_f64<..., SEMI>) replayed with the device's
// pass structure: passes of `loge` stages, the last one partial; inside a pass a butterfly's outputs are reduced only when the next
// stage adds them (bit t/2 of the index clear), the last stage of every pass reduces everything
static double g_semi_max = 0, g_semi_inner = 0;
// uni_only (round 5, ntt_core_f64.hpp SEMIU): the semi-strict schedule in the wave-uniform passes only -- the first pass and every pass whose
// lanes of a wave share the twiddle group (LO = logn - (pass + 1) loge >= 6) --, plain strict butterflies in the others
static void test_semi(uint64_t n, int loge, uint64_t p, bool adversarial, bool uni_only = false) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    std::vector<uint64_t> blk(4 * n);
    orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
    const uint64_t* roots = blk.data() + 2 * n;
    auto centre = [&](uint64_t v) { return v > p / 2 ? (double)v - (double)p : (double)v; };
    int logn = 0; while ((1ull << logn) < n) ++logn;
    const int passes = (logn + loge - 1) / loge, kl = logn - (passes - 1) * loge;
    std::vector<uint64_t> x(n), ref;
    orc_fill_splitmix(x.data(), n, p ^ (n + 31), p);
    if (adversarial) for (uint64_t i = 0; i < n; ++i) x[i] = (i & 1) ? p / 2 + 1 : ((i & 2) ? p / 2 : p - 1);
    ref = x; orc_ks_ntt(ref.data(), n, p, roots);
    std::vector<double> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = hxf::reduce(hxf::to_f64(x[i]), m);
    int s = 1;
    for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1, ++s) {
        const int pass = (s - 1) / loge, u = (s - 1) % loge, K = pass == passes - 1 ? kl : loge;
        const bool last_in_pass = u == K - 1;
        const bool semi_pass = !uni_only || (pass < passes - 1 && (pass == 0 || logn - (pass + 1) * loge >= 6));
        for (uint64_t i = 0; i < mm; ++i) {
            const double w = centre(roots[mm + i]), wp = w / (double)p;
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                if (!semi_pass) {
                    hxf::ct_bfly(v[j], v[j + t], w, m);
                    for (double q : {v[j], v[j + t]}) { const double a = q < 0 ? -q : q; if (a > g_semi_max) g_semi_max = a; }
                    continue;
                }
                const bool added_next = last_in_pass || ((j & (t >> 1)) == 0);
                {   // intermediates
                    const double h = v[j + t] * w, k = __builtin_rint(v[j + t] * wp), inner = __builtin_fma(-k, m.p, h);
                    const double tt = hxf::mul_shoup(v[j + t], w, wp, m);
                    for (double q : {inner, tt, v[j] + tt, v[j] - tt}) { const double a = q < 0 ? -q : q; if (a > g_semi_inner) g_semi_inner = a; }
                }
                hxf::ct_bfly_semi(v[j], v[j + t], w, wp, m, added_next);
                for (double q : {v[j], v[j + t]}) { const double a = q < 0 ? -q : q; if (a > g_semi_max) g_semi_max = a; }
            }
        }
    }
    for (uint64_t i = 0; i < n; ++i) CHECK(hxf::from_f64_53(hxf::lift(v[i], m)) == ref[i], "semi fwd n=%lu loge=%d p=%lu i=%lu", n, loge, p, i);
}

// the products inside mul_mod / mul_shoup at such a modulus: |h - k p| must stay below 2^53 (exactness of the fma)
static void test_prime_wide(uint64_t p) {
    hxf::Mod m{(double)p, 1.0 / (double)p};
    const int64_t P = (int64_t)p;
    for (int it = 0; it < 400000; ++it) {
        const bool edge = (it & 3) == 0;
        auto pick = [&](int64_t bound) -> int64_t {
            if (edge) { const int64_t e[6] = {bound, -bound, bound - 1, -(bound - 1), bound / 2, 1}; return e[rnd() % 6]; }
            return (int64_t)(rnd() % (2 * (uint64_t)bound + 1)) - bound;
        };
        // forward: |y| <= p/2 + 2, |w| <= p/2;   inverse: |d| <= p + 4
        for (int64_t bound : {P / 2 + 2, P + 4}) {
            const int64_t a = pick(bound), w = pick(P / 2);
            const double h = (double)a * (double)w, k = __builtin_rint(h * m.pinv);
            const double hk = __builtin_fma(-k, m.p, h);
            trackw(hk);
            const double u = hxf::mul_mod((double)a, (double)w, m);
            CHECK(u == (double)(int64_t)u && centred((i128)a * w - (int64_t)u, P) == 0, "wide mul_mod p=%lu a=%ld w=%ld u=%.0f", p, a, w, u);
            trackw(u);
            const double r = hxf::reduce(u, m);
            CHECK((int64_t)r >= -(P / 2) - 2 && (int64_t)r <= P / 2 + 2 && centred((i128)a * w - (int64_t)r, P) == 0, "wide reduce p=%lu", p);
        }
        const int64_t c = pick(P / 2 + 2);
        CHECK(hxf::lift((double)c, m) == (double)centred(c, P), "wide lift p=%lu c=%ld", p, c);
        const uint64_t raw = rnd() % ((uint64_t)1 << 53);
        const double rr = hxf::reduce(hxf::to_f64(raw), m);
        CHECK((int64_t)rr >= -(P / 2) - 2 && (int64_t)rr <= P / 2 + 2 && centred((i128)raw - (int64_t)rr, P) == 0, "wide input reduce p=%lu raw=%lu", p, raw);
    }
}

// ---- round 5: plans whose limbs differ in tier ------------------------------------------------------------------
// Every transform of the keyswitch runs modulo ONE q_i, and since round 5 in the tier of THAT modulus (hexl_ks_plan::tier) instead of the
// tier of the plan's largest one. What is new on the arithmetic side is which values meet which modulus: c_d, canonical modulo a 52-bit
// q_d, range-reduced by a 27-bit q_i (a quotient of 25 bits out of one rounded multiply) and then transformed at reduction period 12;
// the centred remainder of a 27-bit special prime entering a strict 52-bit limb, and of a 52-bit one entering a 27-bit limb. This replays,
// for every ordered pair (q_d -> q_i) of a chain, the non-SKIP mod-up round of keyswitch_x.hip / keyswitch_f64.hip / keyswitch_lat.hip --
// reduce, forward transform in q_i's tier without the reduction after the last stage, multiply-accumulate in q_i's form (folded with
// the accumulator at its bound; strict limbs: the strict fold) -- and the mod-down input, against the oracle and 128-bit integers.
static double g_mixed_max = 0;
static inline void trackm(double x) { double a = x < 0 ? -x : x; if (a > g_mixed_max) g_mixed_max = a; }
static void test_mixed_chain(uint64_t n, const std::vector<uint64_t>& chain, bool adversarial) {
    int logn = 0; while ((1ull << logn) < n) ++logn;
    for (size_t ii = 0; ii < chain.size(); ++ii) {
        const uint64_t p = chain[ii];
        const hxf::Mod m{(double)p, 1.0 / (double)p};
        const int period = hxf::lazy_period_for((double)p);          // 0 = strict: what hexl_ks_plan_create gives limb ii
        std::vector<uint64_t> blk(4 * n);
        orc_tables_keyswitch(n, p, orc_minimal_primitive_root(2 * n, p), blk.data());
        const uint64_t* roots = blk.data() + 2 * n;
        auto centre = [&](uint64_t v) { return hxf::reduce(hxf::to_f64(v), m); };
        std::vector<uint64_t> key(n);
        orc_fill_splitmix(key.data(), n, p ^ 0x2718, p);
        for (size_t dd = 0; dd <= chain.size(); ++dd) {
            if (dd == ii) continue;
            // dd < size: c_d canonical modulo q_d; dd == size: y = s' - floor(q_sp/2) of EVERY other modulus as special prime (centred)
            std::vector<double> u(n);
            std::vector<uint64_t> ref(n);
            const uint64_t qd = dd < chain.size() ? chain[dd] : chain[(ii + 1) % chain.size()];
            for (uint64_t i = 0; i < n; ++i) {
                double in;
                if (dd < chain.size()) {
                    const uint64_t x = adversarial ? qd - 1 - (i & 3) : rnd() % qd;
                    in = hxf::to_f64_lt52(x);
                    ref[i] = x % p;
                } else {
                    const int64_t y = adversarial ? ((i & 1) ? 1 : -1) * (int64_t)(qd / 2) : (int64_t)(rnd() % qd) - (int64_t)(qd / 2);
                    in = (double)y;
                    ref[i] = (uint64_t)centred((i128)y, (int64_t)p);
                }
                u[i] = hxf::reduce(in, m);                                   // intt1_redu.hpp:36-42 / intt2_redu.hpp:49-51 (non-SKIP kernels)
                trackm(in); trackm(u[i]);
                CHECK((u[i] < 0 ? -u[i] : u[i]) <= 0.5 * (double)p + 2.0, "mixed reduce q_d=%lu -> p=%lu i=%lu: %.0f", qd, p, i, u[i]);
            }
            orc_ks_ntt(ref.data(), n, p, roots);
            int s = 1;
            for (uint64_t mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1, ++s) {
                const bool red = period == 0 || hxf::lazy_fwd_reduce_after(s, 0, period);       // FINAL = false: no reduction after the last stage
                for (uint64_t i = 0; i < mm; ++i) {
                    const double w = centre(roots[mm + i]);
                    for (uint64_t j = 2 * i * t; j < 2 * i * t + t; ++j) {
                        if (red) hxf::ct_bfly(u[j], u[j + t], w, m); else hxf::ct_bfly_lazy(u[j], u[j + t], w, m);
                        trackm(u[j]); trackm(u[j + t]);
                    }
                }
            }
            for (uint64_t i = 0; i < n; ++i) {
                CHECK(hxf::from_f64(hxf::lift(hxf::reduce(u[i], m), m)) == ref[i], "mixed fwd q_d=%lu -> p=%lu (period %d) i=%lu", qd, p, period, i);
                const uint64_t kv = adversarial ? ((i & 1) ? p / 2 : p / 2 + 1) : key[i];
                const double kc = centre(kv);
                const double bound = period == 0 ? 0.9 : 1.6;                // accumulator bounds of mac_fold's two tiers (f64_arith.hpp)
                const double acc0 = ((u[i] < 0) != (kc < 0) ? -1.0 : 1.0) * (double)(uint64_t)(bound * (double)p);
                const double acc = hxf::mac_fold(acc0, u[i], kc, m);
                trackm(acc);
                const i128 exact = (i128)(int64_t)acc0 + (i128)(int64_t)u[i] * (int64_t)kc;
                CHECK(acc == (double)(int64_t)acc && centred(exact - (int64_t)acc, (int64_t)p) == 0 &&
                      (acc < 0 ? -acc : acc) <= (period == 0 ? 0.9 : 1.7) * (double)p + 2.0,
                      "mixed mac_fold q_d=%lu -> p=%lu i=%lu acc=%.0f", qd, p, i, acc);
            }
        }
    }
}

int main() {
    std::vector<uint64_t> primes;
    uint64_t tmp[8];
    orc_generate_primes(tmp, 8, 51, 16384);                   // the keyswitch bench primes, just above 2^51
    for (int i = 0; i < 8; ++i) primes.push_back(tmp[i]);
    for (uint64_t v = (1ull << 52) - 32767; primes.size() < 11 && v > (1ull << 51); v -= 32768)   // largest < 2^52, = 1 mod 2^15
        if (orc_is_prime(v)) primes.push_back(v);
    orc_generate_primes(tmp, 2, 30, 16384); primes.push_back(tmp[0]);
    orc_generate_primes(tmp, 2, 16, 1024);  primes.push_back(tmp[0]);   // ~2^16, the API's lower bound
    orc_generate_primes(tmp, 2, 40, 16384); primes.push_back(tmp[1]);
    for (uint64_t p : primes) test_prime(p);
    for (uint64_t p : primes) {
        test_transforms(1024, p);
        if (p > (1ull << 50)) test_transforms(16384, p);
    }
    // lazy schedules: only moduli <= LAZY_MAX_MODULUS; include the largest admissible prime = 1 mod 2^15
    {
        std::vector<uint64_t> lazy_primes(primes.begin(), primes.begin() + 8);
        for (uint64_t v = ((uint64_t)hxf::LAZY_MAX_MODULUS / 32768) * 32768 + 1; v > (1ull << 51); v -= 32768)
            if (orc_is_prime(v)) { lazy_primes.push_back(v); break; }
        lazy_primes.push_back(primes[primes.size() - 3]);      // a 30-bit prime
        for (uint64_t p : lazy_primes) {
            CHECK((double)p <= hxf::LAZY_MAX_MODULUS, "prime %lu not lazy-admissible", p);
            for (uint64_t n : {1024ull, 2048ull, 16384ull}) { test_transforms_lazy(n, p, false); test_transforms_lazy(n, p, true); }
        }
        for (uint64_t p : primes) if ((double)p <= hxf::LAZY_MAX_MODULUS) { test_mac_fold(p, 2.14); test_mac_fold(p, 3.46); }
        // round 4: shifted schedule / un-reduced accumulators / table-free inverse, at the ratio bound and at equal-sized moduli
        for (uint64_t p : lazy_primes)
            for (uint64_t n : {1024ull, 2048ull, 4096ull, 8192ull, 16384ull})
                for (int adv = 0; adv < 2; ++adv) {
                    if ((double)p > (double)(1ull << 50)) { test_round4(n, p, 3, hxf::LAZY_SKIP_MAX_RATIO, adv); test_round4(n, p, 3, 1.008, adv); }
                    test_inverse_nowp(n, p, true, adv);
                    test_inverse_nowp(n, p, true, adv, hxf::LAZY_SKIP_MAX_RATIO);       // standalone _INTT fast path (ntt.hip fast_path_limit)
                }
        std::printf("lazy schedules: max |x| seen = 2^%.3f (limit 2^53)\n", log2(g_max_abs));
        CHECK(g_max_abs < 9007199254740992.0, "lazy bound exceeded");
        // longer reduction periods for smaller moduli (f64_arith.hpp: lazy_period_for): the largest admissible prime
        // = 1 mod 2^15 of each tier and one well inside it, at the period the plan would choose
        for (int tier = 0; tier < 2; ++tier) {
            const uint64_t top = tier == 0 ? (1ull << 50) : (1ull << 49);
            const int period = tier == 0 ? 6 : 12;
            std::vector<uint64_t> tp;
            for (uint64_t v = top - 32767; tp.empty(); v -= 32768) if (orc_is_prime(v)) tp.push_back(v);
            orc_generate_primes(tmp, 2, tier == 0 ? 49 : 47, 16384); tp.push_back(tmp[0]);
            g_max_abs = 0;
            for (uint64_t p : tp) {
                CHECK(hxf::lazy_period_for((double)p) >= period, "prime %lu: period %d not admissible", p, period);
                for (uint64_t n : {1024ull, 16384ull}) { test_transforms_lazy(n, p, false, period); test_transforms_lazy(n, p, true, period); }
            }
            for (uint64_t p : tp) test_mac_fold(p, tier == 0 ? 4.82 : 9.5);      // two un-reduced tail stages after a reduction
            for (uint64_t p : tp) test_mac_fold(p, tier == 0 ? 6.3 : 11.9);      // a whole period un-reduced (shifted schedule)
            for (uint64_t p : tp)
                for (uint64_t n : {1024ull, 16384ull})
                    for (int adv = 0; adv < 2; ++adv) { test_round4(n, p, period, hxf::LAZY_SKIP_MAX_RATIO, adv); test_inverse_nowp(n, p, true, adv); }
            std::printf("period %2d (p <= 2^%d): max |x| seen = 2^%.3f (limit 2^53)\n", period, tier == 0 ? 50 : 49, log2(g_max_abs));
            CHECK(g_max_abs < 9007199254740992.0, "lazy bound exceeded for period %d", period);
        }
        // round 6: the X schedules, every tier at its largest admissible prime = 1 mod 2^15, the bench primes and one prime well inside the tier;
        // every transform size of the kernels, both input kinds, both consumers
        for (int period : {3, 6, 12}) {
            const uint64_t top = period == 3 ? (uint64_t)hxf::LAZY_MAX_MODULUS : period == 6 ? (1ull << 50) : (1ull << 49);
            std::vector<uint64_t> tp;
            for (uint64_t v = ((top - 1) / 32768) * 32768 + 1; tp.empty(); v -= 32768) if (v <= top && orc_is_prime(v)) tp.push_back(v);
            if (period == 3) { tp.push_back(primes[0]); tp.push_back(primes[7]); }
            orc_generate_primes(tmp, 2, period == 3 ? 50 : period == 6 ? 49 : 47, 16384); tp.push_back(tmp[0]);
            g_xs_max = 0; g_xs_tail = 0;
            for (uint64_t p : tp) {
                CHECK(hxf::lazy_period_for((double)p) >= period, "prime %lu: tier %d not admissible", p, period);
                for (uint64_t n : {1024ull, 2048ull, 4096ull, 8192ull, 16384ull, 32768ull}) {
                    if (p % (2 * n) != 1) continue;                          // (15 stages: primes = 1 mod 2^16 only)
                    for (int adv = 0; adv < 2; ++adv)
                        for (int shift = 0; shift < 2; ++shift)
                            for (int down = 0; down < 2; ++down) {
                                test_xsched(n, p, period, shift, down, hxf::LAZY_SKIP_MAX_RATIO, adv);
                                if (n == 16384) test_xsched(n, p, period, shift, down, 1.004, adv);
                            }
                }
            }
            g_is_max = 0;
            for (uint64_t p : tp)
                for (auto ge : {std::pair<uint64_t, int>{1024, 4}, {2048, 4}, {2048, 5}, {4096, 4}, {4096, 5}, {8192, 4}, {8192, 5}, {16384, 4}})
                    for (int adv = 0; adv < 2; ++adv) {
                        test_isched(ge.first, ge.second, p, period, adv, hxf::LAZY_SKIP_MAX_RATIO);
                        test_isched(ge.first, ge.second, p, period, adv, 1.0);
                    }
            std::printf("I schedules, period-%d tier (%d primes): max |x| seen = 2^%.3f (limit 2^53)\n", period, (int)tp.size(), log2(g_is_max));
            CHECK(g_is_max < 9007199254740992.0, "I schedule bound exceeded (tier %d)", period);
            std::printf("X schedules, period-%d tier (%d primes): max |x| seen = 2^%.3f (limit 2^53), largest tail %.3f p\n", period, (int)tp.size(), log2(g_xs_max), g_xs_tail);
            CHECK(g_xs_max < 9007199254740992.0, "X schedule bound exceeded (tier %d)", period);
        }
    }
    // strict kernels (moduli up to 2^52): table-free inverse with both outputs reduced
    for (uint64_t p : primes)
        for (uint64_t n : {1024ull, 16384ull}) { test_inverse_nowp(n, p, false, false); test_inverse_nowp(n, p, false, true); }
    // strict tier of the folded multiply-accumulate: every prime above the lazy bound, the largest 52-bit ones among them
    {
        int n_strict = 0;
        for (uint64_t p : primes) if ((double)p > hxf::LAZY_MAX_MODULUS) { test_mac_fold_strict(p); ++n_strict; }
        CHECK(n_strict >= 2, "no strict primes in the list");
        std::printf("strict folded multiply-accumulate (%d primes): max |acc| / p = %.3f (bound 0.9), largest intermediate 2^%.3f (limit 2^53)\n",
                    n_strict, g_fold_strict_max, log2(g_fold_strict_inner));
        CHECK(g_fold_strict_inner < 9007199254740992.0, "strict mac_fold intermediate");
    }
    // semi-strict forward schedule: the largest 52-bit primes, SURVEY 8d's 2^52 + 393217 (inside SEMI_MAX_MODULUS), a mid-range strict
    // prime; every geometry the kernels use (16 or 32 coefficients per thread)
    {
        std::vector<uint64_t> sp;
        for (uint64_t p : primes) if ((double)p > hxf::LAZY_MAX_MODULUS) sp.push_back(p);
        sp.push_back(4503599627763713ull);
        for (uint64_t v = (3ull << 50) + 1; sp.size() < 6; v += 32768) if (orc_is_prime(v)) sp.push_back(v);      // ~2^51.58
        for (uint64_t p : sp) {
            CHECK((double)p <= hxf::SEMI_MAX_MODULUS && (double)p > hxf::LAZY_MAX_MODULUS, "semi prime %lu", p);
            for (int adv = 0; adv < 2; ++adv) {
                test_semi(16384, 4, p, adv); test_semi(16384, 5, p, adv); test_semi(1024, 4, p, adv); test_semi(8192, 5, p, adv);
                test_semi(4096, 4, p, adv); test_semi(2048, 4, p, adv);
                test_semi(16384, 4, p, adv, true); test_semi(8192, 4, p, adv, true); test_semi(1024, 4, p, adv, true);   // strict keyswitch kernels (SEMIU)
            }
        }
        std::printf("semi-strict forward schedule (%d primes): max |x| after a stage = 2^%.3f, largest intermediate 2^%.3f (limit 2^53)\n",
                    (int)sp.size(), log2(g_semi_max), log2(g_semi_inner));
        CHECK(g_semi_inner < 9007199254740992.0 && g_semi_max < 9007199254740992.0, "semi-strict bound exceeded");
    }
    // strict kernels above 2^52 (standalone _NTT / _INTT only): SURVEY 8d's prime and the largest admissible one = 1 mod 2^15
    {
        std::vector<uint64_t> wide = {4503599627763713ull};
        for (uint64_t v = ((hxf::STRICT_NTT_MAX_Q - 1) / 32768) * 32768 + 1; wide.size() < 2 && v > (1ull << 52); v -= 32768)
            if (v < hxf::STRICT_NTT_MAX_Q && orc_is_prime(v)) wide.push_back(v);
        for (uint64_t p : wide) {
            CHECK(orc_is_prime(p) && p % 32768 == 1 && p >= (1ull << 52) && p < hxf::STRICT_NTT_MAX_Q, "wide prime %lu", p);
            test_prime_wide(p);
            for (uint64_t n : {1024ull, 16384ull}) { test_strict_wide(n, p, false); test_strict_wide(n, p, true); }
        }
        std::printf("strict kernels at 2^52 <= p < 2^52 * 1.125 (%lu, %lu): max |x| seen = 2^%.3f (limit 2^53)\n", wide[0], wide.back(), log2(g_wide_max));
        CHECK(g_wide_max < 9007199254740992.0, "strict wide bound exceeded");
    }
    // round 5, limbs of different tiers: bridge-seal's chain (largest primes = 1 mod 2n below 2^52, 2^30, 2^30, 2^40, 2^27, 2^27, 2^27;
    // seal_test.sh:20) and a ladder through all four tiers, every ordered pair of moduli
    {
        for (uint64_t n : {1024ull, 16384ull}) {
            std::vector<uint64_t> seal, ladder;
            for (int b : {52, 30, 30, 40, 27, 27, 27})
                for (uint64_t v = (1ull << b) - 2 * n + 1;; v -= 2 * n) {
                    bool used = false;
                    for (uint64_t q : seal) used = used || q == v;
                    if (!used && orc_is_prime(v)) { seal.push_back(v); break; }
                }
            for (int b : {52, 51, 50, 49, 44})
                for (uint64_t v = (1ull << b) - 2 * n + 1;; v -= 2 * n)
                    if (orc_is_prime(v)) { ladder.push_back(v); break; }
            if (n == 16384) CHECK(seal[0] == 4503599626682369ull && seal[6] == 132612097ull, "seal chain primes %lu %lu", seal[0], seal[6]);
            CHECK(hxf::lazy_period_for((double)seal[0]) == 0 && hxf::lazy_period_for((double)seal[3]) == 12 && hxf::lazy_period_for((double)seal[6]) == 12,
                  "seal chain tiers");
            CHECK(hxf::lazy_period_for((double)ladder[1]) == 3 && hxf::lazy_period_for((double)ladder[2]) == 6 && hxf::lazy_period_for((double)ladder[3]) == 12,
                  "ladder tiers");
            for (int adv = 0; adv < 2; ++adv) { test_mixed_chain(n, seal, adv); test_mixed_chain(n, ladder, adv); }
        }
        std::printf("limbs of different tiers (seal chain 52,30,30,40,27,27,27 + four-tier ladder, every pair): max |x| seen = 2^%.3f (limit 2^53)\n",
                    log2(g_mixed_max));
        CHECK(g_mixed_max < 9007199254740992.0, "mixed-tier bound exceeded");
    }
    std::printf("folded multiply-accumulate: max |acc| / p seen = %.3f (bound 1.6)\n", g_fold_max);
    std::printf(failures ? "F64 SELFTEST: %d FAILURE(S)\n" : "F64 SELFTEST: ALL PASSED (%d primes)\n", failures ? failures : (int)primes.size());
    return failures ? 1 : 0;
}
