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
 * Not vectorised by hand (no AVX-512 IFMA like HEXL's production kernels): kind = "port".
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
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
} cb_mod;

struct cb_plan {
    uint64_t n, L, K;
    cb_mod* m;                     /* [K] */
    const uint64_t* const* keys;   /* caller-owned: keys[d][(k*K+i)*n+j] */
    uint64_t** key_p;              /* Shoup factors of the keys: [d][(k*(L+1)+slot)*n+j] */
};

static inline uint64_t mulhi(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }
/* x*w mod q in [0, 2q) for x < 2^64, w < q, wp = floor(w*2^64/q) */
static inline uint64_t mul_shoup_lazy(uint64_t x, uint64_t w, uint64_t wp, uint64_t q) { return w * x - mulhi(x, wp) * q; }
static inline uint64_t csub(uint64_t x, uint64_t q) { return x >= q ? x - q : x; }
/* x mod q for x < 2^64 */
static inline uint64_t barrett1(uint64_t x, const cb_mod* m) { return csub(x - mulhi(x, m->barr_hi) * m->q, m->q); }

/* Harvey forward NTT, inputs < q (or < 4q), outputs fully reduced; tests/test_utils/ntt.cpp:474-548 */
static void fwd_ntt(uint64_t* x, uint64_t n, const cb_mod* m) {
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
static void inv_ntt(uint64_t* x, uint64_t n, const cb_mod* m) {
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
        orc_tables_keyswitch(n, q, orc_minimal_primitive_root(2 * n, q), blk);     /* the oracle's (= reference's) tables */
        m->w = (uint64_t*)malloc(n * 8); m->wp = (uint64_t*)malloc(n * 8);
        m->iw = (uint64_t*)malloc(n * 8); m->iwp = (uint64_t*)malloc(n * 8);
        memcpy(m->iw, blk, n * 8);
        memcpy(m->w, blk + 2 * n, n * 8);
        for (uint64_t j = 0; j < n; j++) { m->wp[j] = orc_shoup_factor(m->w[j], q); m->iwp[j] = orc_shoup_factor(m->iw[j], q); }
        m->inv_n = orc_invmod(n, q); m->inv_n_p = orc_shoup_factor(m->inv_n, q);
        m->msf = modswitch[i] % q; m->msf_p = orc_shoup_factor(m->msf, q);
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
                for (uint64_t j = 0; j < n; j++) kp[j] = orc_shoup_factor(key[j] % p->m[i].q, p->m[i].q);
            }
    }
    return p;
}

void cb_plan_destroy(struct cb_plan* p) {
    if (!p) return;
    for (uint64_t i = 0; i < p->K; i++) { free(p->m[i].w); free(p->m[i].wp); free(p->m[i].iw); free(p->m[i].iwp); }
    for (uint64_t d = 0; d < p->L; d++) free(p->key_p[d]);
    free(p->key_p); free(p->m); free(p);
}

/* one keyswitch, SURVEY 2.1-K4 steps 1-7, result accumulated into; `ws` = (L + 2(L+1) + 2) * n words of scratch */
static void cb_one(const struct cb_plan* p, uint64_t* result, const uint64_t* t_target, uint64_t* ws) {
    const uint64_t n = p->n, L = p->L, K = p->K, sp = K - 1;
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
            else { for (uint64_t j = 0; j < n; j++) u[j] = barrett1(src[j], m); fwd_ntt(u, n, m); }
            for (uint64_t k = 0; k < 2; k++) {
                const uint64_t* key = p->keys[d] + (k * K + i) * n;
                const uint64_t* kp = p->key_p[d] + (k * (L + 1) + slot) * n;
                uint64_t* pr = prod + (k * (L + 1) + slot) * n;
                for (uint64_t j = 0; j < n; j++)                      /* lazy: prod < 2q throughout */
                    pr[j] = csub(pr[j] + mul_shoup_lazy(u[j], key[j], kp[j], m->q), m->twoq);
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
            for (uint64_t j = 0; j < n; j++) u[j] = barrett1(s[j] + m->fix, m);
            fwd_ntt(u, n, m);
            const uint64_t* pr = prod + (k * (L + 1) + i) * n;
            uint64_t* res = result + (k * L + i) * n;
            for (uint64_t j = 0; j < n; j++) {
                const uint64_t in = csub(pr[j], m->q) + m->q - u[j];                 /* < 2q */
                const uint64_t out = csub(mul_shoup_lazy(in, m->msf, m->msf_p, m->q), m->q);
                res[j] = csub(res[j] + out, m->q);
            }
        }
    }
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
