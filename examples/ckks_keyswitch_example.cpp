// ckks_keyswitch_example.cpp -- end-to-end use of the public API (include/hexl-fpga.h) on an RLWE key switch,
// standing in for the reference's SEAL test (experimental/bridge-seal/tests/keyswitch-example.cpp:119-206, which
// needs SEAL + HEXL): build real switching keys s_new -> s_old with a special prime, switch a random polynomial
// with intel::hexl::KeySwitch on the GPU, and check  result0 + result1*s_old == t*s_new + small noise  in every
// RNS limb. All transforms are done with the library's own _NTT / _INTT, the products with plain host arithmetic.
//   make -C examples && ./examples/ckks_keyswitch_example
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/hexl-fpga.h"
#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
using namespace intel::hexl;
typedef unsigned __int128 u128;
typedef std::vector<uint64_t> vec;

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
static uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) { uint64_t r = 1; for (b %= q; e; e >>= 1) { if (e & 1) r = mulmod(r, b, q); b = mulmod(b, b, q); } return r; }
static uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); }
static bool is_prime(uint64_t n) {
    for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) { if (n == a) return true; if (n % a == 0) return false; }
    uint64_t d = n - 1; int r = 0; while (!(d & 1)) { d >>= 1; ++r; }
    for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        uint64_t x = powmod(a, d, n); if (x == 1 || x == n - 1) continue;
        bool ok = false; for (int i = 1; i < r && !ok; ++i) { x = mulmod(x, x, n); ok = x == n - 1; }
        if (!ok) return false;
    }
    return true;
}
static uint64_t bitrev(uint64_t x, int bits) { uint64_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

struct Modulus {                     // tables in the layouts _NTT / _INTT expect (HEXL layouts)
    uint64_t q, n, inv_n, inv_n_w;
    vec roots, precon, iroots, iprecon;
    Modulus(uint64_t q_, uint64_t n_, int logn) : q(q_), n(n_), roots(n_), precon(n_), iroots(n_), iprecon(n_) {
        uint64_t w = 0;                                   // minimal primitive 2n-th root (what KeySwitch derives itself)
        for (uint64_t g = 2; !w; ++g) { uint64_t c = powmod(g, (q - 1) / (2 * n), q); if (powmod(c, n, q) == q - 1) w = c; }
        { uint64_t sq = mulmod(w, w, q), cur = w, best = w; for (uint64_t i = 0; i < n; ++i) { if (cur < best) best = cur; cur = mulmod(cur, sq, q); } w = best; }
        vec pre(n);
        roots[0] = 1; pre[0] = 1; uint64_t prev = 0;
        for (uint64_t i = 1; i < n; ++i) { uint64_t idx = bitrev(i, logn); roots[idx] = mulmod(roots[prev], w, q); pre[idx] = invmod(roots[idx], q); prev = idx; }
        iroots[0] = 1; uint64_t pos = 1;
        for (uint64_t m = n >> 1; m > 0; m >>= 1) for (uint64_t i = 0; i < m; ++i) iroots[pos++] = pre[m + i];
        for (uint64_t i = 0; i < n; ++i) { precon[i] = (uint64_t)(((u128)roots[i] << 64) / q); iprecon[i] = (uint64_t)(((u128)iroots[i] << 64) / q); }
        inv_n = invmod(n, q); inv_n_w = mulmod(inv_n, iroots[n - 1], q);
    }
    void ntt(vec& x) const { _NTT(x.data(), roots.data(), precon.data(), q, n); }
    void intt(vec& x) const { _INTT(x.data(), iroots.data(), iprecon.data(), q, inv_n, inv_n_w, n); }
};

static uint64_t rng_s = 12345;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

int main() {
    const uint64_t n = 8192; const int logn = 13; const uint64_t L = 3, K = 4;
    vec primes;
    for (uint64_t v = (1ull << 50) + 1; primes.size() < K; v += 2 * n) if (is_prime(v)) primes.push_back(v);
    acquire_FPGA_resources();
    std::vector<Modulus> mod;
    for (uint64_t q : primes) mod.emplace_back(q, n, logn);
    const uint64_t P = primes[K - 1];
    auto lift_small = [&](const std::vector<int>& s, uint64_t q) { vec r(n); for (uint64_t j = 0; j < n; ++j) r[j] = s[j] < 0 ? q - (uint64_t)(-s[j]) : (uint64_t)s[j]; return r; };
    std::vector<int> s_old(n), s_new(n);
    for (auto& v : s_old) v = (int)(rnd() % 3) - 1;
    for (auto& v : s_new) v = (int)(rnd() % 3) - 1;
    // switching keys: for decomposition index d, limb i:  b = -a*s_old + e + (i == d ? P : 0)*s_new   (NTT domain)
    std::vector<vec> keys(L, vec(2 * K * n));
    for (uint64_t d = 0; d < L; ++d) {
        std::vector<int> e(n); for (auto& v : e) v = (int)(rnd() % 7) - 3;
        vec a_int(n); for (auto& v : a_int) v = rnd() >> 2;
        for (uint64_t i = 0; i < K; ++i) {
            const uint64_t q = primes[i];
            vec a(n), so = lift_small(s_old, q), sn = lift_small(s_new, q), ee = lift_small(e, q);
            for (uint64_t j = 0; j < n; ++j) a[j] = a_int[j] % q;
            mod[i].ntt(a); mod[i].ntt(so); mod[i].ntt(sn); mod[i].ntt(ee);
            for (uint64_t j = 0; j < n; ++j) {
                uint64_t b = (q - mulmod(a[j], so[j], q) + ee[j]) % q;
                if (i == d) b = (b + mulmod(P % q, sn[j], q)) % q;
                keys[d][(0 * K + i) * n + j] = b;
                keys[d][(1 * K + i) * n + j] = a[j];
            }
        }
    }
    std::vector<const uint64_t*> key_ptrs; for (auto& k : keys) key_ptrs.push_back(k.data());
    vec moduli(primes), msf(K, 1);
    for (uint64_t i = 0; i + 1 < K; ++i) msf[i] = invmod(P % primes[i], primes[i]);
    // the polynomial to switch: one integer polynomial, given in NTT form per limb
    vec t_int(n); for (auto& v : t_int) v = rnd() >> 2;
    vec t(L * n), result(2 * L * n, 0);
    for (uint64_t d = 0; d < L; ++d) { vec x(n); for (uint64_t j = 0; j < n; ++j) x[j] = t_int[j] % primes[d]; mod[d].ntt(x); for (uint64_t j = 0; j < n; ++j) t[d * n + j] = x[j]; }

    KeySwitch(result.data(), t.data(), n, L, K, L + 1, 2, moduli.data(), key_ptrs.data(), msf.data());

    long long worst = 0; bool consistent = true; std::vector<long long> ref_noise(n);
    for (uint64_t i = 0; i < L; ++i) {
        const uint64_t q = primes[i];
        vec so = lift_small(s_old, q), sn = lift_small(s_new, q), lhs(n);
        mod[i].ntt(so); mod[i].ntt(sn);
        for (uint64_t j = 0; j < n; ++j) {
            uint64_t v = (result[(0 * L + i) * n + j] + mulmod(result[(1 * L + i) * n + j], so[j], q)) % q;
            lhs[j] = (v + q - mulmod(t[i * n + j], sn[j], q)) % q;
        }
        mod[i].intt(lhs);
        for (uint64_t j = 0; j < n; ++j) {
            long long c = lhs[j] <= q / 2 ? (long long)lhs[j] : (long long)lhs[j] - (long long)q;
            if (llabs(c) > worst) worst = llabs(c);
            if (i == 0) ref_noise[j] = c; else consistent &= (ref_noise[j] == c);
        }
    }
    release_FPGA_resources();
    std::printf("key switch of a degree-%lu polynomial over %lu x 50-bit limbs: max |noise| = %lld (limbs %s)\n", n, L, worst,
                consistent ? "agree" : "DISAGREE");
    const bool ok = consistent && worst < (1ll << 26);
    std::printf(ok ? "EXAMPLE PASSED\n" : "EXAMPLE FAILED\n");
    return ok ? 0 : 1;
}
