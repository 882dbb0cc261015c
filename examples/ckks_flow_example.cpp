// ckks_flow_example.cpp -- a mini-CKKS evaluator on the public API (include/hexl-fpga.h), standing in for the reference's
// SEAL bridge test (experimental/bridge-seal/tests/keyswitch-example.cpp:119-206, which needs SEAL 4.0 + HEXL 1.2.4):
//
//     encode -> encrypt -> multiply -> relinearize -> rescale -> rotate by one slot -> decrypt -> decode
//
// with that test's parameters (experimental/bridge-seal/tests/seal_test.sh:20): N = 16384, coefficient modulus bit sizes
// 52,30,30,40,27,27,27 (the last prime is the key-switching special prime), scale 2^52, random inputs in
// (-2^13, 2^13), every slot within 5e-5 of x[i+1]^2. What runs on the GPU through the library is exactly what SEAL's
// hexl-fpga bridge offloads or could offload: every forward / inverse transform (_NTT / _INTT), the ciphertext product
// (DyadicMultiply) and both key switches (KeySwitch: relinearisation at 6 decomposition limbs, the Galois key switch at 5
// -- mixed 27..52-bit moduli, worksize 1, the way the bridge drives it). Sampling, the canonical embedding (a complex FFT)
// and the few element-wise products of encryption / rescaling / decryption are plain host code, as they are in SEAL.
//
//     ckks_flow_example [log2 N = 14] [loops = 1]         exit code 0 and "EXAMPLE PASSED" when every slot is within 5e-5
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hexl-fpga.h"
#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
using namespace intel::hexl;
typedef unsigned __int128 u128;
typedef std::vector<uint64_t> vec;
typedef long double real;
typedef std::complex<real> cplx;

// ---------------------------------------------------------------------------------------------- number theory (host)
static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
static uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { const uint64_t s = a + b; return s >= q ? s - q : s; }
static uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
static uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1;
    for (b %= q; e; e >>= 1) { if (e & 1) r = mulmod(r, b, q); b = mulmod(b, b, q); }
    return r;
}
static uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); }
static bool is_prime(uint64_t n) {
    static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (uint64_t a : bases) { if (n == a) return true; if (n % a == 0) return false; }
    uint64_t d = n - 1; int r = 0;
    while (!(d & 1)) { d >>= 1; ++r; }
    for (uint64_t a : bases) {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool ok = false;
        for (int i = 1; i < r && !ok; ++i) { x = mulmod(x, x, n); ok = x == n - 1; }
        if (!ok) return false;
    }
    return true;
}
static uint64_t bitrev(uint64_t x, int bits) { uint64_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

// the largest primes below 2^bits that are 1 mod 2n, distinct from those already taken (what seal::CoeffModulus::Create picks)
static uint64_t next_prime_below(int bits, uint64_t n, const vec& taken) {
    for (uint64_t v = ((1ull << bits) - 1) / (2 * n) * (2 * n) + 1;; v -= 2 * n)
        if (v < (1ull << bits) && is_prime(v) && std::find(taken.begin(), taken.end(), v) == taken.end()) return v;
}

struct Modulus {   // one RNS prime with the tables _NTT / _INTT expect (HEXL layouts; the root KeySwitch derives for itself)
    uint64_t q, n, inv_n, inv_n_w;
    vec roots, precon, iroots, iprecon;
    Modulus(uint64_t q_, uint64_t n_, int logn) : q(q_), n(n_), roots(n_), precon(n_), iroots(n_), iprecon(n_) {
        uint64_t w = 0;                                   // minimal primitive 2n-th root of unity
        for (uint64_t g = 2; !w; ++g) { const uint64_t c = powmod(g, (q - 1) / (2 * n), q); if (powmod(c, n, q) == q - 1) w = c; }
        { const uint64_t sq = mulmod(w, w, q); uint64_t cur = w, best = w; for (uint64_t i = 0; i < n; ++i) { if (cur < best) best = cur; cur = mulmod(cur, sq, q); } w = best; }
        vec pre(n);
        roots[0] = 1; pre[0] = 1;
        uint64_t prev = 0;
        for (uint64_t i = 1; i < n; ++i) { const uint64_t idx = bitrev(i, logn); roots[idx] = mulmod(roots[prev], w, q); pre[idx] = invmod(roots[idx], q); prev = idx; }
        iroots[0] = 1;
        uint64_t pos = 1;
        for (uint64_t m = n >> 1; m > 0; m >>= 1) for (uint64_t i = 0; i < m; ++i) iroots[pos++] = pre[m + i];
        for (uint64_t i = 0; i < n; ++i) { precon[i] = (uint64_t)(((u128)roots[i] << 64) / q); iprecon[i] = (uint64_t)(((u128)iroots[i] << 64) / q); }
        inv_n = invmod(n, q); inv_n_w = mulmod(inv_n, iroots[n - 1], q);
    }
};

// ---------------------------------------------------------------------------------------------- the evaluator
struct Ckks {
    uint64_t n; int logn; size_t K;                 // K primes: K - 1 data primes + the special prime (last)
    std::vector<Modulus> mod;
    vec q;
    real scale;
    std::vector<int> s;                             // ternary secret
    std::vector<vec> relin, galois;                 // switching keys, [d][(k*K + i)*n + j] (host/src/fpga.cpp:1186-1190)
    vec msf;                                        // P^-1 mod q_i
    uint64_t rng = 0x9E3779B97F4A7C15ull;

    uint64_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
    int noise() { int v = 0; for (int i = 0; i < 12; ++i) v += (int)(rnd() & 1); return v - 6; }   // centred binomial, sigma ~ 1.7

    // transforms of several limbs in one worksize window each (limb k of `xs` is modulo mod[idx[k]])
    void ntt(std::vector<vec*> xs, const std::vector<size_t>& idx) {
        _set_worksize_NTT(xs.size());
        for (size_t k = 0; k < xs.size(); ++k) { const Modulus& m = mod[idx[k]]; _NTT(xs[k]->data(), m.roots.data(), m.precon.data(), m.q, n); }
        _NTTCompleted();
    }
    void intt(std::vector<vec*> xs, const std::vector<size_t>& idx) {
        _set_worksize_INTT(xs.size());
        for (size_t k = 0; k < xs.size(); ++k) { const Modulus& m = mod[idx[k]]; _INTT(xs[k]->data(), m.iroots.data(), m.iprecon.data(), m.q, m.inv_n, m.inv_n_w, n); }
        _INTTCompleted();
    }
    vec small_ntt(const std::vector<int>& p, size_t i) {      // a small signed polynomial in NTT form modulo q_i
        vec r(n);
        for (uint64_t j = 0; j < n; ++j) r[j] = p[j] < 0 ? q[i] - (uint64_t)(-p[j]) : (uint64_t)p[j];
        ntt({&r}, {i});
        return r;
    }
    std::vector<int> automorph(const std::vector<int>& p, uint64_t g) {     // p(X) -> p(X^g), X^n = -1
        std::vector<int> r(n);
        for (uint64_t j = 0; j < n; ++j) { const uint64_t e = j * g % (2 * n); if (e < n) r[e] = p[j]; else r[e - n] = -p[j]; }
        return r;
    }
    // switching key from `target` (NTT form per prime) to s: key d, limb i = (b, a), b = -a s + e + [i == d] P target
    std::vector<vec> make_switch_key(const std::vector<vec>& target_ntt, const std::vector<vec>& s_ntt) {
        const size_t L = K - 1;
        std::vector<vec> keys(L, vec(2 * K * n));
        for (size_t d = 0; d < L; ++d) {
            std::vector<int> e(n);
            for (auto& v : e) v = noise();
            for (size_t i = 0; i < K; ++i) {
                vec a(n), en = small_ntt(e, i);
                for (auto& v : a) v = rnd() % q[i];               // uniform residues per prime = a uniform element mod Q P
                const uint64_t P = q[K - 1] % q[i];
                for (uint64_t j = 0; j < n; ++j) {
                    uint64_t b = addmod(submod(0, mulmod(a[j], s_ntt[i][j], q[i]), q[i]), en[j], q[i]);
                    if (i == d) b = addmod(b, mulmod(P, target_ntt[i][j], q[i]), q[i]);
                    keys[d][(0 * K + i) * n + j] = b;
                    keys[d][(1 * K + i) * n + j] = a[j];
                }
            }
        }
        return keys;
    }

    Ckks(int logn_, const std::vector<int>& bits, real scale_) : n(1ull << logn_), logn(logn_), K(bits.size()), scale(scale_) {
        for (int b : bits) q.push_back(next_prime_below(b, n, q));
        for (uint64_t p : q) mod.emplace_back(p, n, logn);
        s.resize(n);
        for (auto& v : s) v = (int)(rnd() % 3) - 1;
        std::vector<vec> s_ntt(K), s2_ntt(K), sg_ntt(K);
        const std::vector<int> sg = automorph(s, 5);
        for (size_t i = 0; i < K; ++i) {
            s_ntt[i] = small_ntt(s, i);
            sg_ntt[i] = small_ntt(sg, i);
            s2_ntt[i].resize(n);
            for (uint64_t j = 0; j < n; ++j) s2_ntt[i][j] = mulmod(s_ntt[i][j], s_ntt[i][j], q[i]);
        }
        relin = make_switch_key(s2_ntt, s_ntt);             // s^2 -> s
        galois = make_switch_key(sg_ntt, s_ntt);            // s(X^5) -> s  (rotation by one slot)
        msf.assign(K, 1);
        for (size_t i = 0; i + 1 < K; ++i) msf[i] = invmod(q[K - 1] % q[i], q[i]);
    }

    // ---- canonical embedding: slot i <-> evaluation at zeta^(5^i), zeta = exp(i pi / n) -------------------------------
    static void fft(std::vector<cplx>& a, bool inverse) {
        const size_t m = a.size();
        for (size_t i = 1, j = 0; i < m; ++i) { size_t bit = m >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(a[i], a[j]); }
        const real pi = acosl(-1.0L);
        for (size_t len = 2; len <= m; len <<= 1) {
            const real ang = 2 * pi / (real)len * (inverse ? -1 : 1);
            std::vector<cplx> w(len / 2);
            for (size_t k = 0; k < len / 2; ++k) w[k] = cplx(cosl(ang * k), sinl(ang * k));
            for (size_t i = 0; i < m; i += len)
                for (size_t k = 0; k < len / 2; ++k) { const cplx u = a[i + k], v = a[i + k + len / 2] * w[k]; a[i + k] = u + v; a[i + k + len / 2] = u - v; }
        }
        if (inverse) for (auto& x : a) x /= (real)m;
    }
    // real coefficients of the polynomial whose value at zeta^(5^i) is x[i] (and the conjugate at the conjugate point)
    std::vector<real> embed_inverse(const std::vector<double>& x) {
        std::vector<cplx> V(n);
        uint64_t g = 1;
        for (uint64_t i = 0; i < n / 2; ++i, g = g * 5 % (2 * n)) { V[(g - 1) / 2] = cplx(x[i], 0); V[(2 * n - g - 1) / 2] = cplx(x[i], 0); }
        fft(V, true);                                       // b_j = (1/n) sum_k V[k] omega^(-jk): p(zeta^(2k+1)) = sum_j (a_j zeta^j) omega^(jk)
        const real pi = acosl(-1.0L);
        std::vector<real> a(n);
        for (uint64_t j = 0; j < n; ++j) a[j] = (V[j] * cplx(cosl(pi * j / n), -sinl(pi * j / n))).real();
        return a;
    }
    std::vector<double> embed(const std::vector<real>& a) {
        const real pi = acosl(-1.0L);
        std::vector<cplx> b(n);
        for (uint64_t j = 0; j < n; ++j) b[j] = a[j] * cplx(cosl(pi * j / n), sinl(pi * j / n));
        fft(b, false);
        std::vector<double> x(n / 2);
        uint64_t g = 1;
        for (uint64_t i = 0; i < n / 2; ++i, g = g * 5 % (2 * n)) x[i] = (double)b[(g - 1) / 2].real();
        return x;
    }

    // ciphertext: [component][limb][n] flattened, NTT form, `L` data limbs, under s, at `scale`
    struct Ct { vec c; size_t comps, L; real scale; };

    Ct encrypt(const std::vector<double>& x) {
        const size_t L = K - 1;
        std::vector<real> a = embed_inverse(x);
        std::vector<int> e(n);
        for (auto& v : e) v = noise();
        Ct ct{vec(2 * L * n), 2, L, scale};
        std::vector<vec> m(L, vec(n)), c1(L, vec(n));
        for (size_t i = 0; i < L; ++i) {
            for (uint64_t j = 0; j < n; ++j) {                // round(scale * a_j) + e_j, reduced modulo q_i (|.| < 2^70)
                const __int128 v = (__int128)roundl(a[j] * scale) + e[j];
                const __int128 r = v % (__int128)q[i];
                m[i][j] = (uint64_t)(r < 0 ? r + q[i] : r);
                c1[i][j] = rnd() % q[i];
            }
        }
        std::vector<vec*> ptrs; std::vector<size_t> idx;
        for (size_t i = 0; i < L; ++i) { ptrs.push_back(&m[i]); idx.push_back(i); }
        ntt(ptrs, idx);
        for (size_t i = 0; i < L; ++i) {
            const vec sn = small_ntt(s, i);
            for (uint64_t j = 0; j < n; ++j) {                // c0 = m + e - c1 s   (c1 is sampled straight in NTT form)
                ct.c[(0 * L + i) * n + j] = submod(m[i][j], mulmod(c1[i][j], sn[j], q[i]), q[i]);
                ct.c[(1 * L + i) * n + j] = c1[i][j];
            }
        }
        return ct;
    }

    // wall time of the two KeySwitch call sites (worksize 1, the SEAL bridge's shape: the call returns when the result is there)
    // (the first call of a site also builds the device plan of its key set -- tables, key upload -- and is kept apart)
    double ks_first_us[2] = {0, 0}, ks_us[2] = {0, 0};
    int ks_calls[2] = {0, 0};
    void note_ks(int site, double us) { if (ks_calls[site]++ == 0) ks_first_us[site] = us; else ks_us[site] += us; }
    // (a0, a1) x (b0, b1) -> (a0 b0, a0 b1 + a1 b0, a1 b1): intel::hexl::DyadicMultiply on the GPU
    Ct multiply(const Ct& a, const Ct& b) {
        Ct r{vec(3 * a.L * n), 3, a.L, a.scale * b.scale};
        vec moduli(q.begin(), q.begin() + a.L);
        DyadicMultiply(r.c.data(), a.c.data(), b.c.data(), n, moduli.data(), a.L);
        return r;
    }
    // (c0, c1, c2) -> (c0, c1) + KeySwitch_{s^2 -> s}(c2): intel::hexl::KeySwitch accumulates into its result argument
    Ct relinearize(const Ct& a) {
        const size_t L = a.L;
        Ct r{vec(a.c.begin(), a.c.begin() + 2 * L * n), 2, L, a.scale};
        std::vector<const uint64_t*> kp;
        for (size_t d = 0; d < L; ++d) kp.push_back(relin[d].data());
        const auto t0 = std::chrono::steady_clock::now();
        KeySwitch(r.c.data(), a.c.data() + 2 * L * n, n, L, K, L + 1, 2, q.data(), kp.data(), msf.data());
        note_ks(0, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        return r;
    }
    // divide by the last data prime with rounding and drop it (SEAL's rescale_to_next)
    Ct rescale(const Ct& a) {
        const size_t L = a.L, l = L - 1;
        const uint64_t ql = q[l], half = ql >> 1;
        Ct r{vec(a.comps * l * n), a.comps, l, a.scale / (real)ql};
        for (size_t c = 0; c < a.comps; ++c) {
            vec last(a.c.begin() + (c * L + l) * n, a.c.begin() + (c * L + l + 1) * n);
            intt({&last}, {l});
            std::vector<vec> t(l, vec(n));
            std::vector<vec*> ptrs; std::vector<size_t> idx;
            for (size_t i = 0; i < l; ++i) {
                for (uint64_t j = 0; j < n; ++j)             // (last + half) mod q_l, brought to q_i, minus half: the centred remainder
                    t[i][j] = submod(addmod(last[j], half, ql) % q[i], half % q[i], q[i]);
                ptrs.push_back(&t[i]); idx.push_back(i);
            }
            ntt(ptrs, idx);
            for (size_t i = 0; i < l; ++i) {
                const uint64_t inv = invmod(ql % q[i], q[i]);
                for (uint64_t j = 0; j < n; ++j)
                    r.c[(c * l + i) * n + j] = mulmod(submod(a.c[(c * L + i) * n + j], t[i][j], q[i]), inv, q[i]);
            }
        }
        return r;
    }
    // slots move one place to the left: X -> X^5 on both components, then KeySwitch_{s(X^5) -> s}
    Ct rotate_by_one(const Ct& a) {
        const size_t L = a.L;
        std::vector<vec> limb(2 * L, vec(n));
        std::vector<vec*> ptrs; std::vector<size_t> idx;
        for (size_t x = 0; x < 2 * L; ++x) { std::copy(a.c.begin() + x * n, a.c.begin() + (x + 1) * n, limb[x].begin()); ptrs.push_back(&limb[x]); idx.push_back(x % L); }
        intt(ptrs, idx);
        for (size_t x = 0; x < 2 * L; ++x) {
            const uint64_t qi = q[x % L];
            vec r(n);
            for (uint64_t j = 0; j < n; ++j) { const uint64_t e = j * 5 % (2 * n); if (e < n) r[e] = limb[x][j]; else r[e - n] = submod(0, limb[x][j], qi); }
            limb[x] = r;
        }
        ntt(ptrs, idx);
        Ct r{vec(2 * L * n, 0), 2, L, a.scale};
        vec t(L * n);
        for (size_t i = 0; i < L; ++i) {
            std::copy(limb[i].begin(), limb[i].end(), r.c.begin() + i * n);          // c0(X^5); component 1 starts at zero
            std::copy(limb[L + i].begin(), limb[L + i].end(), t.begin() + i * n);    // c1(X^5) is what gets switched
        }
        std::vector<const uint64_t*> kp;
        for (size_t d = 0; d < L; ++d) kp.push_back(galois[d].data());
        const auto t0 = std::chrono::steady_clock::now();
        KeySwitch(r.c.data(), t.data(), n, L, K, L + 1, 2, q.data(), kp.data(), msf.data());
        note_ks(1, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        return r;
    }
    // c0 + c1 s, back to coefficients, CRT-composed and centred exactly (256-bit integers), divided by the scale
    std::vector<double> decrypt(const Ct& a) {
        const size_t L = a.L;
        std::vector<vec> m(L, vec(n));
        std::vector<vec*> ptrs; std::vector<size_t> idx;
        for (size_t i = 0; i < L; ++i) {
            const vec sn = small_ntt(s, i);
            for (uint64_t j = 0; j < n; ++j) m[i][j] = addmod(a.c[(0 * L + i) * n + j], mulmod(a.c[(1 * L + i) * n + j], sn[j], q[i]), q[i]);
            ptrs.push_back(&m[i]); idx.push_back(i);
        }
        intt(ptrs, idx);
        // Garner: x = v_0 + v_1 q_0 + v_2 q_0 q_1 + ..., 0 <= v_i < q_i
        std::vector<vec> inv(L, vec(L, 0));
        for (size_t i = 0; i < L; ++i) for (size_t k = 0; k < i; ++k) inv[i][k] = invmod(q[k] % q[i], q[i]);
        struct U256 { uint64_t w[4]; };
        auto mul_add = [](U256 x, uint64_t m_, uint64_t add) { u128 c = add; for (int k = 0; k < 4; ++k) { c += (u128)x.w[k] * m_; x.w[k] = (uint64_t)c; c >>= 64; } return x; };
        auto geq = [](const U256& x, const U256& y) { for (int k = 3; k >= 0; --k) if (x.w[k] != y.w[k]) return x.w[k] > y.w[k]; return true; };
        auto sub = [](const U256& x, const U256& y) { U256 r; u128 br = 0; for (int k = 0; k < 4; ++k) { const u128 d = (u128)x.w[k] - y.w[k] - br; r.w[k] = (uint64_t)d; br = (d >> 64) & 1; } return r; };
        auto to_real = [](const U256& x) { real r = 0; for (int k = 3; k >= 0; --k) r = r * 18446744073709551616.0L + (real)x.w[k]; return r; };
        U256 Q{{1, 0, 0, 0}};
        for (size_t i = 0; i < L; ++i) Q = mul_add(Q, q[i], 0);
        U256 halfQ = Q;
        for (int k = 0; k < 4; ++k) halfQ.w[k] = (Q.w[k] >> 1) | (k < 3 ? Q.w[k + 1] << 63 : 0);
        std::vector<real> coeff(n);
        std::vector<uint64_t> v(L);
        for (uint64_t j = 0; j < n; ++j) {
            for (size_t i = 0; i < L; ++i) {
                uint64_t t = m[i][j] % q[i];
                for (size_t k = 0; k < i; ++k) t = mulmod(submod(t, v[k] % q[i], q[i]), inv[i][k], q[i]);
                v[i] = t;
            }
            U256 x{{0, 0, 0, 0}};
            for (size_t i = L; i-- > 0;) x = mul_add(x, q[i], v[i]);      // Horner over the mixed-radix digits
            coeff[j] = geq(x, halfQ) ? -to_real(sub(Q, x)) : to_real(x);
            coeff[j] /= a.scale;
        }
        return embed(coeff);
    }
};

int main(int argc, char** argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 14;
    const int loops = argc > 2 ? atoi(argv[2]) : 1;
    if (logn < 10 || logn > 14) { std::printf("log2 N must be between 10 and 14 (the reference's KeySwitch sizes)\n"); return 2; }
    const std::vector<int> bits = {52, 30, 30, 40, 27, 27, 27};         // seal_test.sh:20; the last one is the special prime
    const double precision = 0.00005, bound = 8192.0;                   // keyswitch-example.cpp: test_precision, 2^(27/2)
    acquire_FPGA_resources();
    Ckks ck(logn, bits, ldexpl(1.0L, 52));
    std::printf("mini-CKKS on hexl-fpga (MI355X): N = %lu, coefficient modulus", (unsigned long)ck.n);
    for (uint64_t p : ck.q) std::printf(" %lu", (unsigned long)p);
    std::printf(" (bits 52,30,30,40,27,27,27; special prime last), scale 2^52\n");
    bool all_ok = true;
    for (int loop = 0; loop < loops; ++loop) {
        std::vector<double> x(ck.n / 2);
        for (auto& v : x) v = ((double)(ck.rnd() >> 11) / 9007199254740992.0 * 2 - 1) * bound;
        Ckks::Ct ct = ck.encrypt(x);
        ct = ck.multiply(ct, ct);                    // DyadicMultiply
        ct = ck.relinearize(ct);                     // KeySwitch, 6 decomposition limbs
        ct = ck.rescale(ct);                         // _INTT / _NTT
        ct = ck.rotate_by_one(ct);                   // _INTT / _NTT, KeySwitch with the Galois key, 5 decomposition limbs
        const std::vector<double> out = ck.decrypt(ct);
        double worst = 0;
        for (size_t i = 0; i < x.size(); ++i) {
            const double want = x[(i + 1) % x.size()] * x[(i + 1) % x.size()];
            worst = std::max(worst, std::fabs(out[i] - want));
        }
        const bool ok = worst < precision;
        all_ok = all_ok && ok;
        std::printf("loop %d: encrypt -> multiply -> relinearize -> rescale -> rotate -> decrypt: max |error| over %zu slots = %.3g (%s, tolerance %.0e)\n",
                    loop, x.size(), worst, ok ? "SUCCESS" : "FAIL", precision);
    }
    if (ck.ks_calls[0] > 1 && ck.ks_calls[1] > 1)
        std::printf("KeySwitch end to end at worksize 1 (host pointers, PCIe included): relinearize (6 limbs) %.0f us, rotate (5 limbs) %.0f us "
                    "on average over %d calls each; first calls (plan + key upload) %.1f / %.1f ms\n", ck.ks_us[0] / (ck.ks_calls[0] - 1),
                    ck.ks_us[1] / (ck.ks_calls[1] - 1), ck.ks_calls[0] - 1, ck.ks_first_us[0] / 1e3, ck.ks_first_us[1] / 1e3);
    release_FPGA_resources();
    std::printf(all_ok ? "EXAMPLE PASSED\n" : "EXAMPLE FAILED\n");
    return all_ok ? 0 : 1;
}
