/*
 * cpu_baseline.c -- the CPU leg that bench.py times next to the MI355X (TEST/BENCH INFRASTRUCTURE ONLY, like the rest of
 * oracle/). The north star asks for "the Intel HEXL CPU path timed on the box's own host cores"; HEXL (v1.2.4,
 * cmake/intel-hexl/intel-hexl.cmake:6-7) is not vendored and there is no network, so this is a PORT of the same
 * algorithms the reference's RUN_CHOICE=0 path reaches (host/src/fpga_int.cpp:473-477 -> intel::hexl::internal::
 * KeySwitch): Harvey lazy butterflies with Shoup-preconditioned twiddles (the scheme of tests/test_utils/ntt.cpp:474-659,
 * which restates HEXL's NTT), Barrett reduction for the key products, tables precomputed once per parameter set, and
 * OpenMP over independent ciphertexts (SURVEY 8d "CPU baseline in the same run"). It must return exactly what
 * orc_keyswitch returns (tests/test_cpu_baseline.py) -- only faster: ~6x one thread of the line-by-line oracle, which
 * divides 128-bit products and rebuilds its tables on every call.
 *
 * Vector code, selected at run time like HEXL selects its kernels (its NTT: AVX512-IFMA with 52-bit Shoup factors for moduli
 * below 2^50, AVX512-DQ with 64-bit lanes and an emulated 64x64 high product above, scalar otherwise): the same two kernel
 * families are written out below with intrinsics (butterflies on 8 lanes; the three narrow stages through lane permutes;
 * the element-wise mod-up / multiply-accumulate / mod-down loops as well) when the compiler targets a host that has them
 * (-march=native on the box that runs the benchmark). cb_isa() says which one a plan uses; HEXL_CPU_ISA=scalar forces the
 * scalar port. Every path is lazy inside and fully reduced at the end, so all of them return the oracle's words.
 * kind = "port+avx512ifma" / "port+avx512dq" / "port" in the bench line.
 */
#define _GNU_SOURCE
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "hexl_oracle.h"

typedef unsigned __int128 u128;

typedef struct {
    uint64_t q, twoq, barr_hi;     /* barr_hi = floor(2^64 / q) for single-word Barrett */
    uint64_t inv_n, inv_n_p;       /* n^-1 and its Shoup factor */
    uint64_t msf, msf_p, half_mod, fix;
    uint64_t *w, *wp;              /* forward roots (bit-reversed, index m+i) + Shoup factors */
    uint64_t *iw, *iwp;            /* inverse roots in stage order from index 0 + Shoup factors */
    int isa;                       /* 0 scalar, 1 AVX512-DQ 64-bit lanes, 2 AVX512-IFMA (q < 2^50): all *_p factors are then */
                                   /* floor(. * 2^52 / q) instead of floor(. * 2^64 / q) */
} cb_mod;

/* everything a keyswitch READS besides its own ciphertext: per-modulus constants + tables, keys, key Shoup factors. The plan
 * holds one view over the caller's / its own arrays; cb_keyswitch_timed adds one replica per NUMA node its threads run on
 * (round 4: with a single copy first-touched by one thread, 128 threads pulled 29 MB per keyswitch out of one node's memory:
 * 3.9 % parallel efficiency). */
struct cb_view {
    cb_mod* m;                     /* [K] */
    const uint64_t* const* keys;   /* keys[d][(k*K+i)*n+j] */
    uint64_t** key_p;              /* Shoup factors of the keys: [d][(k*(L+1)+slot)*n+j] */
};
#define CB_MAX_NODES 16
struct cb_plan {
    uint64_t n, L, K;
    cb_mod* m;                     /* [K] */
    const uint64_t* const* keys;   /* caller-owned: keys[d][(k*K+i)*n+j] */
    uint64_t** key_p;              /* Shoup factors of the keys: [d][(k*(L+1)+slot)*n+j] */
    int nrep;                      /* NUMA replicas made so far */
    int rep_node[CB_MAX_NODES];
    struct cb_view rep[CB_MAX_NODES];
};

static inline uint64_t mulhi(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }
/* x*w mod q in [0, 2q) for x < 2^64, w < q, wp = floor(w*2^64/q) */
static inline uint64_t mul_shoup_lazy(uint64_t x, uint64_t w, uint64_t wp, uint64_t q) { return w * x - mulhi(x, wp) * q; }
static inline uint64_t csub(uint64_t x, uint64_t q) { return x >= q ? x - q : x; }
/* x mod q for x < 2^64 */
static inline uint64_t barrett1(uint64_t x, const cb_mod* m) { return csub(x - mulhi(x, m->barr_hi) * m->q, m->q); }

/* Harvey forward NTT, inputs < q (or < 4q), outputs fully reduced; tests/test_utils/ntt.cpp:474-548 */
static void fwd_ntt_scalar(uint64_t* x, uint64_t n, const cb_mod* m) {
    const uint64_t q = m->q, twoq = m->twoq;
    uint64_t t = n >> 1;
    for (uint64_t mm = 1; mm < n; mm <<= 1, t >>= 1)
        for (uint64_t i = 0; i < mm; i++) {
            const uint64_t W = m->w[mm + i], Wp = m->wp[mm + i];
            uint64_t *X = x + 2 * i * t, *Y = X + t;
            for (uint64_t j = 0; j < t; j++) {
                const uint64_t tx = csub(X[j], twoq);
                const uint64_t Q = mul_shoup_lazy(Y[j], W, Wp, q);
                X[j] = tx + Q;
                Y[j] = tx + twoq - Q;
            }
        }
    for (uint64_t j = 0; j < n; j++) x[j] = csub(csub(x[j], twoq), q);
}

/* Harvey inverse NTT (Gentleman-Sande), scaled by n^-1, outputs in [0, q); ntt.cpp:580-659 */
static void inv_ntt_scalar(uint64_t* x, uint64_t n, const cb_mod* m) {
    const uint64_t q = m->q, twoq = m->twoq;
    uint64_t t = 1, acc = 0;
    for (uint64_t mm = n >> 1; mm >= 1; mm >>= 1, t <<= 1) {
        for (uint64_t i = 0; i < mm; i++) {
            const uint64_t W = m->iw[acc + i], Wp = m->iwp[acc + i];
            uint64_t *X = x + 2 * i * t, *Y = X + t;
            for (uint64_t j = 0; j < t; j++) {
                const uint64_t tx = X[j] + Y[j];
                const uint64_t ty = X[j] + twoq - Y[j];
                X[j] = csub(tx, twoq);
                Y[j] = mul_shoup_lazy(ty, W, Wp, q);
            }
        }
        acc += mm;
    }
    for (uint64_t j = 0; j < n; j++) x[j] = csub(mul_shoup_lazy(x[j], m->inv_n, m->inv_n_p, q), q);
}


/* ------------------------------------------------------------------------------------------------ AVX-512 kernels */
#if defined(__AVX512F__) && defined(__AVX512DQ__)
#include <immintrin.h>
#define CB_AVX512 1
#if defined(__AVX512IFMA__)
#define CB_IFMA 1
#endif
typedef __m512i v8;
#define VSET(x) _mm512_set1_epi64((long long)(x))
#define VLD(p) _mm512_loadu_si512((const void*)(p))
#define VST(p, v) _mm512_storeu_si512((void*)(p), (v))
/* x in [0, 2m) -> [0, m): min(x, x - m) as unsigned (x - m wraps above x when x < m) */
static inline v8 v_csub(v8 x, v8 m) { return _mm512_min_epu64(x, _mm512_sub_epi64(x, m)); }
/* high 64 bits of the 64x64 product from four 32x32 products (what HEXL's 64-bit AVX512-DQ kernels do) */
static inline v8 v_mulhi64(v8 a, v8 b) {
    const v8 lo32 = VSET(0xffffffffull);
    const v8 ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
    const v8 w0 = _mm512_mul_epu32(a, b), w1 = _mm512_mul_epu32(a, bh), w2 = _mm512_mul_epu32(ah, b), w3 = _mm512_mul_epu32(ah, bh);
    const v8 s1 = _mm512_add_epi64(w1, _mm512_srli_epi64(w0, 32));
    const v8 s2 = _mm512_add_epi64(w2, _mm512_and_si512(s1, lo32));
    return _mm512_add_epi64(_mm512_add_epi64(w3, _mm512_srli_epi64(s1, 32)), _mm512_srli_epi64(s2, 32));
}
/* x*w mod q in [0, 2q), Shoup form. IFMA: x, w < 2^52, wp = floor(w 2^52 / q), q < 2^50; else wp = floor(w 2^64 / q) */
static inline __attribute__((always_inline)) v8 v_mul_shoup_lazy(v8 x, v8 w, v8 wp, v8 q, const int ifma) {
#ifdef CB_IFMA
    if (ifma) {
        const v8 z = _mm512_setzero_si512();
        const v8 qh = _mm512_madd52hi_epu64(z, x, wp);
        const v8 d = _mm512_sub_epi64(_mm512_madd52lo_epu64(z, x, w), _mm512_madd52lo_epu64(z, qh, q));
        return _mm512_and_si512(d, VSET((1ull << 52) - 1));
    }
#endif
    (void)ifma;
    return _mm512_sub_epi64(_mm512_mullo_epi64(w, x), _mm512_mullo_epi64(v_mulhi64(x, wp), q));
}
/* the three narrow stages pair coefficients inside a block of 16: lane permutes between (A = x[0..7], B = x[8..15]) and (X, Y) */
static const long long PX4[8] = {0, 1, 2, 3, 8, 9, 10, 11}, PY4[8] = {4, 5, 6, 7, 12, 13, 14, 15};
static const long long PX2[8] = {0, 1, 4, 5, 8, 9, 12, 13}, PY2[8] = {2, 3, 6, 7, 10, 11, 14, 15};
static const long long PX1[8] = {0, 2, 4, 6, 8, 10, 12, 14}, PY1[8] = {1, 3, 5, 7, 9, 11, 13, 15};
static const long long QA2[8] = {0, 1, 8, 9, 2, 3, 10, 11}, QB2[8] = {4, 5, 12, 13, 6, 7, 14, 15};
static const long long QA1[8] = {0, 8, 1, 9, 2, 10, 3, 11}, QB1[8] = {4, 12, 5, 13, 6, 14, 7, 15};
static const long long W4[8] = {0, 0, 0, 0, 1, 1, 1, 1}, W2[8] = {0, 0, 1, 1, 2, 2, 3, 3};
/* the 16 / (2t) twiddles of a block spread over the 8 butterfly lanes */
static inline v8 v_twiddles(const uint64_t* w, uint64_t t) {
    if (t == 1) return VLD(w);
    if (t == 2) return _mm512_permutexvar_epi64(VLD(W2), _mm512_castsi256_si512(_mm256_loadu_si256((const __m256i*)w)));
    return _mm512_permutexvar_epi64(VLD(W4), _mm512_castsi128_si512(_mm_loadu_si128((const __m128i*)w)));
}
static inline __attribute__((always_inline)) void v_fwd_ntt(uint64_t* x, uint64_t n, const cb_mod* m, const int ifma) {
    const v8 q = VSET(m->q), twoq = VSET(m->twoq);
    uint64_t t = n >> 1;
    for (uint64_t mm = 1; mm < n; mm <<= 1, t >>= 1) {
        if (t >= 8) {
            for (uint64_t i = 0; i < mm; i++) {
                const v8 W = VSET(m->w[mm + i]), Wp = VSET(m->wp[mm + i]);
                uint64_t *X = x + 2 * i * t, *Y = X + t;
                for (uint64_t j = 0; j < t; j += 8) {
                    const v8 tx = v_csub(VLD(X + j), twoq);
                    const v8 Q = v_mul_shoup_lazy(VLD(Y + j), W, Wp, q, ifma);
                    VST(X + j, _mm512_add_epi64(tx, Q));
                    VST(Y + j, _mm512_sub_epi64(_mm512_add_epi64(tx, twoq), Q));
                }
            }
        } else {
            const v8 px = VLD(t == 4 ? PX4 : t == 2 ? PX2 : PX1), py = VLD(t == 4 ? PY4 : t == 2 ? PY2 : PY1);
            const v8 qa = VLD(t == 4 ? PX4 : t == 2 ? QA2 : QA1), qb = VLD(t == 4 ? PY4 : t == 2 ? QB2 : QB1);
            const uint64_t per = 8 / t;                                   /* twiddles per block of 16 */
            for (uint64_t b = 0; b < n / 16; b++) {
                const v8 A = VLD(x + 16 * b), B = VLD(x + 16 * b + 8);
                const v8 W = v_twiddles(m->w + mm + per * b, t), Wp = v_twiddles(m->wp + mm + per * b, t);
                const v8 tx = v_csub(_mm512_permutex2var_epi64(A, px, B), twoq);
                const v8 Q = v_mul_shoup_lazy(_mm512_permutex2var_epi64(A, py, B), W, Wp, q, ifma);
                const v8 Xn = _mm512_add_epi64(tx, Q), Yn = _mm512_sub_epi64(_mm512_add_epi64(tx, twoq), Q);
                VST(x + 16 * b, _mm512_permutex2var_epi64(Xn, qa, Yn));
                VST(x + 16 * b + 8, _mm512_permutex2var_epi64(Xn, qb, Yn));
            }
        }
    }
    for (uint64_t j = 0; j < n; j += 8) VST(x + j, v_csub(v_csub(VLD(x + j), twoq), q));
}
static inline __attribute__((always_inline)) void v_inv_ntt(uint64_t* x, uint64_t n, const cb_mod* m, const int ifma) {
    const v8 q = VSET(m->q), twoq = VSET(m->twoq);
    uint64_t t = 1, acc = 0;
    for (uint64_t mm = n >> 1; mm >= 1; mm >>= 1, t <<= 1) {
        if (t >= 8) {
            for (uint64_t i = 0; i < mm; i++) {
                const v8 W = VSET(m->iw[acc + i]), Wp = VSET(m->iwp[acc + i]);
                uint64_t *X = x + 2 * i * t, *Y = X + t;
                for (uint64_t j = 0; j < t; j += 8) {
                    const v8 a = VLD(X + j), b = VLD(Y + j);
                    VST(X + j, v_csub(_mm512_add_epi64(a, b), twoq));
                    VST(Y + j, v_mul_shoup_lazy(_mm512_sub_epi64(_mm512_add_epi64(a, twoq), b), W, Wp, q, ifma));
                }
            }
        } else {
            const v8 px = VLD(t == 4 ? PX4 : t == 2 ? PX2 : PX1), py = VLD(t == 4 ? PY4 : t == 2 ? PY2 : PY1);
            const v8 qa = VLD(t == 4 ? PX4 : t == 2 ? QA2 : QA1), qb = VLD(t == 4 ? PY4 : t == 2 ? QB2 : QB1);
            const uint64_t per = 8 / t;
            for (uint64_t b = 0; b < n / 16; b++) {
                const v8 A = VLD(x + 16 * b), B = VLD(x + 16 * b + 8);
                const v8 W = v_twiddles(m->iw + acc + per * b, t), Wp = v_twiddles(m->iwp + acc + per * b, t);
                const v8 xa = _mm512_permutex2var_epi64(A, px, B), ya = _mm512_permutex2var_epi64(A, py, B);
                const v8 Xn = v_csub(_mm512_add_epi64(xa, ya), twoq);
                const v8 Yn = v_mul_shoup_lazy(_mm512_sub_epi64(_mm512_add_epi64(xa, twoq), ya), W, Wp, q, ifma);
                VST(x + 16 * b, _mm512_permutex2var_epi64(Xn, qa, Yn));
                VST(x + 16 * b + 8, _mm512_permutex2var_epi64(Xn, qb, Yn));
            }
        }
        acc += mm;
    }
    const v8 in = VSET(m->inv_n), inp = VSET(m->inv_n_p);
    for (uint64_t j = 0; j < n; j += 8) VST(x + j, v_csub(v_mul_shoup_lazy(VLD(x + j), in, inp, q, ifma), q));
}
/* x mod q for any x < 2^64, barr = floor(2^64 / q) (the mod-up / mod-down operands may exceed 52 bits: always the 64-bit form) */
static inline v8 v_barrett1(v8 x, v8 barr, v8 q) {
    return v_csub(_mm512_sub_epi64(x, _mm512_mullo_epi64(v_mulhi64(x, barr), q)), q);
}
#endif

/* which kernels a modulus gets: HEXL's rule (IFMA below 2^50, 64-bit AVX512-DQ above), HEXL_CPU_ISA=scalar|dq to restrict */
static int pick_isa(uint64_t q, uint64_t n) {
    const char* e = getenv("HEXL_CPU_ISA");
    if (e && !strcmp(e, "scalar")) return 0;
    if (n < 16) return 0;
#ifdef CB_AVX512
    if (!__builtin_cpu_supports("avx512dq")) return 0;
#ifdef CB_IFMA
    if (q < (1ull << 50) && __builtin_cpu_supports("avx512ifma") && !(e && !strcmp(e, "dq"))) return 2;
#endif
    return 1;
#else
    (void)q;
    return 0;
#endif
}
static uint64_t shoup_for(uint64_t y, const cb_mod* m) {     /* floor(y 2^64 / q), or floor(y 2^52 / q) for the IFMA kernels */
    return m->isa == 2 ? (uint64_t)(((u128)y << 52) / m->q) : orc_shoup_factor(y, m->q);
}

struct cb_plan* cb_plan_create(uint64_t n, uint64_t L, uint64_t K, const uint64_t* moduli, const uint64_t* const* keys,
                               const uint64_t* modswitch) {
    struct cb_plan* p = (struct cb_plan*)calloc(1, sizeof(*p));
    p->n = n; p->L = L; p->K = K; p->keys = keys;
    p->m = (cb_mod*)calloc(K, sizeof(cb_mod));
    uint64_t* blk = (uint64_t*)malloc(4 * n * sizeof(uint64_t));
    const uint64_t q_sp = moduli[K - 1];
    for (uint64_t i = 0; i < K; i++) {
        cb_mod* m = &p->m[i];
        const uint64_t q = moduli[i];
        m->q = q; m->twoq = q << 1; m->barr_hi = (uint64_t)(((u128)1 << 64) / q);
        m->isa = pick_isa(q, n);
        orc_tables_keyswitch(n, q, orc_minimal_primitive_root(2 * n, q), blk);     /* the oracle's (= reference's) tables */
        m->w = (uint64_t*)malloc(n * 8); m->wp = (uint64_t*)malloc(n * 8);
        m->iw = (uint64_t*)malloc(n * 8); m->iwp = (uint64_t*)malloc(n * 8);
        memcpy(m->iw, blk, n * 8);
        memcpy(m->w, blk + 2 * n, n * 8);
        for (uint64_t j = 0; j < n; j++) { m->wp[j] = shoup_for(m->w[j], m); m->iwp[j] = shoup_for(m->iw[j], m); }
        m->inv_n = orc_invmod(n, q); m->inv_n_p = shoup_for(m->inv_n, m);
        m->msf = modswitch[i] % q; m->msf_p = shoup_for(m->msf, m);
        m->half_mod = (q_sp >> 1) % q; m->fix = q - m->half_mod;
    }
    free(blk);
    p->key_p = (uint64_t**)calloc(L, sizeof(uint64_t*));
    for (uint64_t d = 0; d < L; d++) {
        p->key_p[d] = (uint64_t*)malloc(2 * (L + 1) * n * 8);
        for (uint64_t k = 0; k < 2; k++)
            for (uint64_t slot = 0; slot <= L; slot++) {
                const uint64_t i = slot < L ? slot : K - 1;
                const uint64_t* key = keys[d] + (k * K + i) * n;
                uint64_t* kp = p->key_p[d] + (k * (L + 1) + slot) * n;
                for (uint64_t j = 0; j < n; j++) kp[j] = shoup_for(key[j] % p->m[i].q, &p->m[i]);
            }
    }
    return p;
}

static void free_view(const struct cb_plan* p, struct cb_view* v) {
    for (uint64_t i = 0; i < p->K; i++) { free(v->m[i].w); free(v->m[i].wp); free(v->m[i].iw); free(v->m[i].iwp); }
    for (uint64_t d = 0; d < p->L; d++) { free((void*)v->keys[d]); free(v->key_p[d]); }
    free(v->m); free((void*)v->keys); free(v->key_p);
}

void cb_plan_destroy(struct cb_plan* p) {
    if (!p) return;
    for (int r = 0; r < p->nrep; r++) free_view(p, &p->rep[r]);
    for (uint64_t i = 0; i < p->K; i++) { free(p->m[i].w); free(p->m[i].wp); free(p->m[i].iw); free(p->m[i].iwp); }
    for (uint64_t d = 0; d < p->L; d++) free(p->key_p[d]);
    free(p->key_p); free(p->m); free(p);
}


static void fwd_ntt(uint64_t* x, uint64_t n, const cb_mod* m) {
#ifdef CB_AVX512
    if (m->isa == 2) { v_fwd_ntt(x, n, m, 1); return; }
    if (m->isa == 1) { v_fwd_ntt(x, n, m, 0); return; }
#endif
    fwd_ntt_scalar(x, n, m);
}
static void inv_ntt(uint64_t* x, uint64_t n, const cb_mod* m) {
#ifdef CB_AVX512
    if (m->isa == 2) { v_inv_ntt(x, n, m, 1); return; }
    if (m->isa == 1) { v_inv_ntt(x, n, m, 0); return; }
#endif
    inv_ntt_scalar(x, n, m);
}
/* the element-wise loops of cb_one, 8 lanes at a time when the modulus has vector kernels */
static void mod_up_reduce(uint64_t* u, const uint64_t* src, uint64_t add, uint64_t n, const cb_mod* m) {      /* u = (src + add) mod q */
#ifdef CB_AVX512
    if (m->isa) {
        const v8 q = VSET(m->q), barr = VSET(m->barr_hi), a = VSET(add);
        for (uint64_t j = 0; j < n; j += 8) VST(u + j, v_barrett1(_mm512_add_epi64(VLD(src + j), a), barr, q));
        return;
    }
#endif
    for (uint64_t j = 0; j < n; j++) u[j] = barrett1(src[j] + add, m);
}
static inline __attribute__((always_inline)) void mac_keys_v(uint64_t* pr, const uint64_t* u, const uint64_t* key, const uint64_t* kp,
                                                             uint64_t n, const cb_mod* m, const int ifma) {
#ifdef CB_AVX512
    const v8 q = VSET(m->q), twoq = VSET(m->twoq);
    for (uint64_t j = 0; j < n; j += 8)
        VST(pr + j, v_csub(_mm512_add_epi64(VLD(pr + j), v_mul_shoup_lazy(VLD(u + j), VLD(key + j), VLD(kp + j), q, ifma)), twoq));
#else
    (void)pr; (void)u; (void)key; (void)kp; (void)n; (void)m; (void)ifma;
#endif
}
static void mac_keys(uint64_t* pr, const uint64_t* u, const uint64_t* key, const uint64_t* kp, uint64_t n, const cb_mod* m) {
    if (m->isa == 2) { mac_keys_v(pr, u, key, kp, n, m, 1); return; }
    if (m->isa == 1) { mac_keys_v(pr, u, key, kp, n, m, 0); return; }
    for (uint64_t j = 0; j < n; j++)                                  /* lazy: prod < 2q throughout */
        pr[j] = csub(pr[j] + mul_shoup_lazy(u[j], key[j], kp[j], m->q), m->twoq);
}
static inline __attribute__((always_inline)) void mod_down_v(uint64_t* res, const uint64_t* pr, const uint64_t* u, uint64_t n,
                                                             const cb_mod* m, const int ifma) {
#ifdef CB_AVX512
    const v8 q = VSET(m->q), msf = VSET(m->msf), msfp = VSET(m->msf_p);
    for (uint64_t j = 0; j < n; j += 8) {
        const v8 in = _mm512_sub_epi64(_mm512_add_epi64(v_csub(VLD(pr + j), q), q), VLD(u + j));               /* < 2q */
        const v8 out = v_csub(v_mul_shoup_lazy(in, msf, msfp, q, ifma), q);
        VST(res + j, v_csub(_mm512_add_epi64(VLD(res + j), out), q));
    }
#else
    (void)res; (void)pr; (void)u; (void)n; (void)m; (void)ifma;
#endif
}
static void mod_down(uint64_t* res, const uint64_t* pr, const uint64_t* u, uint64_t n, const cb_mod* m) {
    if (m->isa == 2) { mod_down_v(res, pr, u, n, m, 1); return; }
    if (m->isa == 1) { mod_down_v(res, pr, u, n, m, 0); return; }
    for (uint64_t j = 0; j < n; j++) {
        const uint64_t in = csub(pr[j], m->q) + m->q - u[j];                 /* < 2q */
        const uint64_t out = csub(mul_shoup_lazy(in, m->msf, m->msf_p, m->q), m->q);
        res[j] = csub(res[j] + out, m->q);
    }
}

/* which kernels the plan's first modulus uses: "scalar", "avx512dq", "avx512ifma" */
const char* cb_isa(const struct cb_plan* p) { return p->m[0].isa == 2 ? "avx512ifma" : p->m[0].isa == 1 ? "avx512dq" : "scalar"; }

/* one keyswitch, SURVEY 2.1-K4 steps 1-7, result accumulated into; `ws` = (L + 2(L+1) + 2) * n words of scratch */
static void cb_one_v(const struct cb_plan* pl, const struct cb_view* p, uint64_t* result, const uint64_t* t_target, uint64_t* ws) {
    const uint64_t n = pl->n, L = pl->L, K = pl->K, sp = K - 1;
    uint64_t *c = ws, *prod = c + L * n, *u = prod + 2 * (L + 1) * n, *s = u + n;
    memset(prod, 0, 2 * (L + 1) * n * 8);
    for (uint64_t d = 0; d < L; d++) {
        memcpy(c + d * n, t_target + d * n, n * 8);
        inv_ntt(c + d * n, n, &p->m[d]);
    }
    for (uint64_t slot = 0; slot <= L; slot++) {
        const uint64_t i = slot < L ? slot : sp;
        const cb_mod* m = &p->m[i];
        for (uint64_t d = 0; d < L; d++) {
            const uint64_t* src = c + d * n;
            if (slot == d) memcpy(u, t_target + d * n, n * 8);       /* NTT(INTT(t_d) mod q_d) = t_d for in-range data */
            else { mod_up_reduce(u, src, 0, n, m); fwd_ntt(u, n, m); }
            for (uint64_t k = 0; k < 2; k++) {
                const uint64_t* key = p->keys[d] + (k * K + i) * n;
                const uint64_t* kp = p->key_p[d] + (k * (L + 1) + slot) * n;
                mac_keys(prod + (k * (L + 1) + slot) * n, u, key, kp, n, m);
            }
        }
    }
    const cb_mod* msp = &p->m[sp];
    for (uint64_t k = 0; k < 2; k++) {
        uint64_t* ps = prod + (k * (L + 1) + L) * n;
        for (uint64_t j = 0; j < n; j++) s[j] = csub(ps[j], msp->q);
        inv_ntt(s, n, msp);
        for (uint64_t j = 0; j < n; j++) s[j] = csub(s[j] + (msp->q >> 1), msp->q);
        for (uint64_t i = 0; i < L; i++) {
            const cb_mod* m = &p->m[i];
            mod_up_reduce(u, s, m->fix, n, m);
            fwd_ntt(u, n, m);
            const uint64_t* pr = prod + (k * (L + 1) + i) * n;
            mod_down(result + (k * L + i) * n, pr, u, n, m);
        }
    }
}

static void cb_one(const struct cb_plan* p, uint64_t* result, const uint64_t* t_target, uint64_t* ws) {
    const struct cb_view v = {p->m, p->keys, p->key_p};
    cb_one_v(p, &v, result, t_target, ws);
}

/* ---- the timed leg: pinned threads, NUMA-local data ------------------------------------------------------------------------ */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* the CPUs this process may use, physical cores first (a CPU that is the first of its SMT sibling list), then the siblings */
static int cpu_order(int* out, int max, int* n_physical) {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set)) return 0;
    int n = 0, nphys = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int c = 0; c < CPU_SETSIZE && n < max; c++) {
            if (!CPU_ISSET(c, &set)) continue;
            char path[128]; int first = c;
            snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
            FILE* f = fopen(path, "r");
            if (f) { if (fscanf(f, "%d", &first) != 1) first = c; fclose(f); }
            const int is_first = (first == c) || !CPU_ISSET(first, &set);
            if ((pass == 0) == is_first) { out[n++] = c; if (pass == 0) nphys++; }
        }
    *n_physical = nphys;
    return n;
}

/* a replica of everything read-only, allocated and first-touched by the calling thread (i.e. on its NUMA node) */
static void make_view(const struct cb_plan* p, struct cb_view* v) {
    const uint64_t n = p->n, L = p->L, K = p->K;
    v->m = (cb_mod*)malloc(K * sizeof(cb_mod));
    memcpy(v->m, p->m, K * sizeof(cb_mod));
    for (uint64_t i = 0; i < K; i++) {
        uint64_t** dst[4] = {&v->m[i].w, &v->m[i].wp, &v->m[i].iw, &v->m[i].iwp};
        uint64_t* src[4] = {p->m[i].w, p->m[i].wp, p->m[i].iw, p->m[i].iwp};
        for (int t = 0; t < 4; t++) { *dst[t] = (uint64_t*)malloc(n * 8); memcpy(*dst[t], src[t], n * 8); }
    }
    uint64_t** keys = (uint64_t**)malloc(L * sizeof(uint64_t*));
    v->key_p = (uint64_t**)malloc(L * sizeof(uint64_t*));
    for (uint64_t d = 0; d < L; d++) {
        keys[d] = (uint64_t*)malloc(2 * K * n * 8);        memcpy(keys[d], p->keys[d], 2 * K * n * 8);
        v->key_p[d] = (uint64_t*)malloc(2 * (L + 1) * n * 8); memcpy(v->key_p[d], p->key_p[d], 2 * (L + 1) * n * 8);
    }
    v->keys = (const uint64_t* const*)keys;
}

/* `threads` OpenMP threads, each PINNED to its own CPU of the process's affinity mask (physical cores first, spread evenly over
 * them when there are fewer threads than cores), each with a private, first-touched copy of the `nsrc` source instances and its
 * own scratch; keys, key Shoup factors and twiddle tables replicated once per NUMA node the threads land on. Every thread walks
 * its instances (result accumulated in place, like the library) until `seconds` have passed. Returns the number of keyswitches
 * done by all threads; *elapsed = the slowest thread's time; *nodes = NUMA nodes used; `first_out` (2 L n words, may be NULL)
 * receives thread 0's first result after ONE keyswitch so that the caller can check this path against the oracle. */
uint64_t cb_keyswitch_timed(struct cb_plan* p, const uint64_t* t_src, const uint64_t* r_src, uint64_t nsrc, int threads,
                            double seconds, double* elapsed, int* nodes, uint64_t* first_out) {
    const uint64_t n = p->n, L = p->L;
    const size_t ws_words = (L + 2 * (L + 1) + 2) * n;
    static int cpus[4096];
    int nphys = 0;
    const int ncpu = cpu_order(cpus, 4096, &nphys);
    uint64_t total = 0;
    double worst = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads) reduction(+ : total) reduction(max : worst)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int t = 0, nt = 1;
#endif
        if (ncpu > 0) {
            /* fewer threads than physical cores: every (nphys / nt)-th core, which spreads them over the sockets and L3 slices */
            const int idx = nt <= nphys ? (int)((long)t * nphys / nt) : t % ncpu;
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[idx], &one);
            sched_setaffinity(0, sizeof(one), &one);
        }
        unsigned cpu = 0, node = 0;
        syscall(SYS_getcpu, &cpu, &node, NULL);
        const struct cb_view* view = NULL;
#pragma omp critical(cb_replicate)
        {
            for (int r = 0; r < p->nrep; r++) if (p->rep_node[r] == (int)node) view = &p->rep[r];
            if (!view && p->nrep < CB_MAX_NODES) {                 /* first thread of this node: allocate + first-touch here */
                make_view(p, &p->rep[p->nrep]);
                p->rep_node[p->nrep] = (int)node;
                view = &p->rep[p->nrep++];
            }
        }
        const struct cb_view fallback = {p->m, p->keys, p->key_p};
        if (!view) view = &fallback;
        uint64_t* ws = (uint64_t*)malloc(ws_words * 8);
        uint64_t* tt = (uint64_t*)malloc(nsrc * L * n * 8);
        uint64_t* rr = (uint64_t*)malloc(nsrc * 2 * L * n * 8);
        memcpy(tt, t_src, nsrc * L * n * 8);
        memcpy(rr, r_src, nsrc * 2 * L * n * 8);
        memset(ws, 0, ws_words * 8);
        cb_one_v(p, view, rr, tt, ws);                              /* warm-up = the checked keyswitch */
        if (t == 0 && first_out) memcpy(first_out, rr, 2 * L * n * 8);
#pragma omp barrier
        const double t0 = now_s();
        uint64_t done = 0;
        for (uint64_t b = 0;; b = (b + 1) % nsrc) {
            cb_one_v(p, view, rr + b * 2 * L * n, tt + b * L * n, ws);
            ++done;
            if (now_s() - t0 >= seconds) break;
        }
        worst = now_s() - t0;
        total = done;
        free(ws); free(tt); free(rr);
        if (ncpu > 0) {                                             /* give the thread back its full mask (the pool is reused) */
            cpu_set_t all;
            CPU_ZERO(&all);
            for (int i = 0; i < ncpu; i++) CPU_SET(cpus[i], &all);
            sched_setaffinity(0, sizeof(all), &all);
        }
    }
    if (elapsed) *elapsed = worst;
    if (nodes) *nodes = p->nrep;
    return total;
}

/* STREAM-style triad (a = b + s c on private, first-touched arrays of `mb` MiB each; 24 bytes moved per element) with the SAME
 * thread placement as cb_keyswitch_timed: what the host's memory system gives this process at `threads` threads, in GB/s -- the
 * yardstick for the keyswitch leg's parallel efficiency (the keys alone are a 29 MB stream per keyswitch and thread). */
double cb_stream_triad(int threads, double seconds, uint64_t mb) {
    static int cpus[4096];
    int nphys = 0;
    const int ncpu = cpu_order(cpus, 4096, &nphys);
    const size_t words = (size_t)mb * (1 << 20) / 8;
    double bytes = 0, worst = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads) reduction(+ : bytes) reduction(max : worst)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int t = 0, nt = 1;
#endif
        if (ncpu > 0) {
            const int idx = nt <= nphys ? (int)((long)t * nphys / nt) : t % ncpu;
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[idx], &one);
            sched_setaffinity(0, sizeof(one), &one);
        }
        double *a = (double*)malloc(words * 8), *b = (double*)malloc(words * 8), *c = (double*)malloc(words * 8);
        for (size_t i = 0; i < words; i++) { a[i] = 0; b[i] = (double)i; c[i] = 1.0; }
#pragma omp barrier
        const double t0 = now_s();
        double moved = 0;
        do {
            for (size_t i = 0; i < words; i++) a[i] = b[i] + 3.0 * c[i];
            __asm__ volatile("" : : "r"(a) : "memory");
            moved += 24.0 * (double)words;
        } while (now_s() - t0 < seconds);
        worst = now_s() - t0;
        bytes = moved;
        free(a); free(b); free(c);
        if (ncpu > 0) {
            cpu_set_t all;
            CPU_ZERO(&all);
            for (int i = 0; i < ncpu; i++) CPU_SET(cpus[i], &all);
            sched_setaffinity(0, sizeof(all), &all);
        }
    }
    return worst > 0 ? bytes / worst / 1e9 : 0.0;
}

/* batch of independent keyswitches, `threads` OpenMP threads (0 = all); returns the number of threads used */
int cb_keyswitch_batch(const struct cb_plan* p, uint64_t* results, const uint64_t* t_targets, uint64_t batch, int threads) {
    const uint64_t n = p->n, L = p->L;
    const size_t ws_words = (L + 2 * (L + 1) + 2) * n;
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel num_threads(threads)
    {
        uint64_t* ws = (uint64_t*)malloc(ws_words * 8);
#pragma omp for schedule(dynamic, 1)
        for (uint64_t b = 0; b < batch; b++) cb_one(p, results + b * 2 * L * n, t_targets + b * L * n, ws);
        free(ws);
    }
#else
    (void)threads;
    uint64_t* ws = (uint64_t*)malloc(ws_words * 8);
    for (uint64_t b = 0; b < batch; b++) cb_one(p, results + b * 2 * L * n, t_targets + b * L * n, ws);
    free(ws);
#endif
    return used;
}

/* batch of forward NTTs (BASELINE metric 2), same scheme */
int cb_ntt_fwd_batch(const struct cb_plan* p, uint64_t modulus_index, uint64_t* x, uint64_t batch, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
    for (uint64_t b = 0; b < batch; b++) fwd_ntt(x + b * p->n, p->n, &p->m[modulus_index]);
    return used;
}

int cb_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
