/*
 * hexl_oracle.c -- CPU restatement of the hexl-fpga hot path. TEST INFRASTRUCTURE ONLY
 * (see hexl_oracle.h). Plain C11 + unsigned __int128, single-threaded, written for
 * clarity: each routine follows the reference loop structure it cites so a reader can
 * diff behaviour line by line. Paths are relative to the intel/hexl-fpga v2.0 tree.
 */
#include "hexl_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ number theory */

/* tests/test_utils/ntt.cpp:51-59 (MultiplyUIntMod = 128-bit product, exact remainder) */
uint64_t orc_mulmod(uint64_t x, uint64_t y, uint64_t q) { return (uint64_t)(((u128)x * y) % q); }

/* tests/test_utils/ntt.cpp:83-95 */
uint64_t orc_powmod(uint64_t base, uint64_t exp, uint64_t q) {
    base %= q;
    uint64_t r = 1;
    while (exp) {
        if (exp & 1) r = orc_mulmod(r, base, q);
        base = orc_mulmod(base, base, q);
        exp >>= 1;
    }
    return r;
}

/* tests/test_utils/ntt.cpp:14-42 -- value only (extended Euclid); result in [0,q) */
uint64_t orc_invmod(uint64_t a, uint64_t q) {
    __int128 t = 0, nt = 1, r = q, nr = a % q;
    while (nr != 0) {
        __int128 qu = r / nr, tmp;
        tmp = t - qu * nt; t = nt; nt = tmp;
        tmp = r - qu * nr; r = nr; nr = tmp;
    }
    if (t < 0) t += q;
    return (uint64_t)t;
}

/* tests/test_utils/ntt.cpp:174-222 (deterministic Miller-Rabin, bases 2..37) */
int orc_is_prime(uint64_t n) {
    static const uint64_t as[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (int i = 0; i < 12; i++) {
        if (n == as[i]) return 1;
        if (n % as[i] == 0) return 0;
    }
    uint64_t d = n - 1;
    unsigned r = 0;
    while ((d & 1) == 0) { d >>= 1; r++; }
    for (int i = 0; i < 12; i++) {
        uint64_t x = orc_powmod(as[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int ok = 0;
        for (unsigned j = 1; j < r; j++) {
            x = orc_mulmod(x, x, n);
            if (x == n - 1) { ok = 1; break; }
        }
        if (!ok) return 0;
    }
    return 1;
}

/* tests/test_utils/ntt.cpp:224-249: candidates 2^bits+1, step 2*ntt_size, below 2^(bits+1) */
size_t orc_generate_primes(uint64_t* out, size_t num, unsigned bits, uint64_t ntt_size) {
    uint64_t v = (1ULL << bits) + 1;
    size_t found = 0;
    while (v < (1ULL << (bits + 1)) && found < num) {
        if (orc_is_prime(v)) out[found++] = v;
        v += 2 * ntt_size;
    }
    return found;
}

/* tests/test_utils/ntt.cpp:111-158 / host/src/number_theory_util.cpp:109-154.
 * The reference draws random candidates; the minimum over all primitive roots is unique,
 * so any deterministic search for one primitive root gives the same answer. */
uint64_t orc_minimal_primitive_root(uint64_t degree, uint64_t q) {
    uint64_t root = 0, e = (q - 1) / degree;
    for (uint64_t g = 2; g < q; g++) {
        uint64_t c = orc_powmod(g, e, q);
        if (c != 0 && orc_powmod(c, degree / 2, q) == q - 1) { root = c; break; }
    }
    uint64_t sq = orc_mulmod(root, root, q), cur = root, best = root;
    for (uint64_t i = 0; i < degree; i++) {
        if (cur < best) best = cur;
        cur = orc_mulmod(cur, sq, q);
    }
    return best;
}

/* tests/test_utils/ntt.cpp:160-171 */
uint64_t orc_reverse_bits(uint64_t x, unsigned width) {
    uint64_t r = 0;
    for (unsigned i = 0; i < width; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* host/inc/number_theory_util.h:197-226 (MultiplyFactor, bit_shift = 64) */
uint64_t orc_shoup_factor(uint64_t op, uint64_t q) { return (uint64_t)(((u128)op << 64) / q); }

static unsigned ilog2(uint64_t n) { unsigned l = 0; while ((1ULL << l) < n) l++; return l; }

/* ------------------------------------------------------------------ twiddle tables */

/* shared first half: roots[bitrev(i)] = roots[bitrev(i-1)] * w; pre[idx] = roots[idx]^-1 */
static void bitrev_powers(uint64_t n, uint64_t q, uint64_t w, uint64_t* roots, uint64_t* inv_pre) {
    unsigned bits = ilog2(n);
    roots[0] = 1; inv_pre[0] = 1;
    uint64_t prev = 0;
    for (uint64_t i = 1; i < n; i++) {
        uint64_t idx = orc_reverse_bits(i, bits);
        roots[idx] = orc_mulmod(roots[prev], w, q);
        inv_pre[idx] = orc_invmod(roots[idx], q);
        prev = idx;
    }
}

/* tests/test_utils/ntt.cpp:290-384 */
void orc_tables_hexl(uint64_t n, uint64_t q, uint64_t w, uint64_t* roots, uint64_t* precon,
                     uint64_t* inv_roots, uint64_t* inv_precon) {
    uint64_t* pre = (uint64_t*)malloc(n * sizeof(uint64_t));
    bitrev_powers(n, q, w, roots, pre);
    inv_roots[0] = pre[0];
    uint64_t pos = 1;                                  /* ntt.cpp:312-324: starts at 1 */
    for (uint64_t m = n >> 1; m > 0; m >>= 1)
        for (uint64_t i = 0; i < m; i++) inv_roots[pos++] = pre[m + i];
    for (uint64_t i = 0; i < n; i++) {
        precon[i] = orc_shoup_factor(roots[i], q);     /* index 0: floor(2^64/q), never read */
        inv_precon[i] = orc_shoup_factor(inv_roots[i], q);
    }
    free(pre);
}

/* host/src/twiddle-factors.cpp:16-62 */
void orc_tables_keyswitch(uint64_t n, uint64_t q, uint64_t w, uint64_t* blk) {
    uint64_t *inv = blk, *inv_precon = blk + n, *roots = blk + 2 * n, *precon = blk + 3 * n;
    uint64_t* pre = (uint64_t*)malloc(n * sizeof(uint64_t));
    bitrev_powers(n, q, w, roots, pre);
    precon[0] = 0;                                     /* twiddle-factors.cpp:40 */
    for (uint64_t i = 1; i < n; i++) precon[i] = orc_shoup_factor(roots[i], q);
    uint64_t pos = 0;                                  /* twiddle-factors.cpp:44-53: from 0 */
    for (uint64_t m = n >> 1; m > 0; m >>= 1)
        for (uint64_t i = 0; i < m; i++) inv[pos++] = pre[m + i];
    inv[n - 1] = 0;                                    /* twiddle-factors.cpp:55 */
    for (uint64_t i = 0; i < n; i++) inv_precon[i] = orc_shoup_factor(inv[i], q);
    free(pre);
}

/* ------------------------------------------------------------------ K1 / K2 */

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }

/* tests/test_utils/ntt.hpp:86-101: W*x - hi64(W'*x)*q, all mod 2^64 */
static inline uint64_t lazy_mul(uint64_t x, uint64_t w, uint64_t wp, uint64_t q) {
    return w * x - mulhi64(x, wp) * q;
}

/* device/fwd_ntt.cpp:137-493; tests/test_utils/ntt.cpp:474-548 */
void orc_ntt_fwd(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* roots, const uint64_t* precon) {
    const uint64_t twoq = q << 1;
    uint64_t t = n >> 1;
    for (uint64_t m = 1; m < n; m <<= 1, t >>= 1) {
        for (uint64_t i = 0; i < m; i++) {
            const uint64_t W = roots[m + i], Wp = precon[m + i];   /* fwd_ntt.cpp:289-291 */
            uint64_t* X = x + 2 * i * t;
            uint64_t* Y = X + t;
            for (uint64_t j = 0; j < t; j++) {
                uint64_t tx = X[j] >= twoq ? X[j] - twoq : X[j];   /* :322-323 */
                uint64_t Q = lazy_mul(Y[j], W, Wp, q);             /* :336-354 */
                X[j] = tx + Q;                                     /* :359 */
                Y[j] = tx + twoq - Q;                              /* :360 */
            }
        }
    }
    for (uint64_t i = 0; i < n; i++) {                             /* :369-384 */
        if (x[i] >= twoq) x[i] -= twoq;
        if (x[i] >= q) x[i] -= q;
    }
}

/* device/inv_ntt.cpp:141-441; tests/test_utils/ntt.cpp:580-659 */
void orc_ntt_inv(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* inv_roots,
                 const uint64_t* inv_precon, uint64_t inv_n, uint64_t inv_n_w) {
    const uint64_t twoq = q << 1;
    uint64_t t = 1, root_index = 1;                                /* inv_ntt.cpp:144 */
    for (uint64_t m = n >> 1; m > 1; m >>= 1, t <<= 1) {
        for (uint64_t i = 0; i < m; i++, root_index++) {
            const uint64_t W = inv_roots[root_index], Wp = inv_precon[root_index];
            uint64_t* X = x + 2 * i * t;
            uint64_t* Y = X + t;
            for (uint64_t j = 0; j < t; j++) {
                uint64_t tx = X[j] + Y[j];                         /* :300-306 */
                uint64_t ty = X[j] + twoq - Y[j];
                X[j] = tx >= twoq ? tx - twoq : tx;
                Y[j] = lazy_mul(ty, W, Wp, q);
            }
        }
    }
    /* last stage fused with n^-1 scaling; Lazy3 computes floor(y*2^64/q) on the fly
     * (device/mod_ops.hpp:135-151; inv_ntt.cpp:400-437) */
    const uint64_t inv_n_p = orc_shoup_factor(inv_n, q), inv_n_w_p = orc_shoup_factor(inv_n_w, q);
    uint64_t* X = x;
    uint64_t* Y = x + (n >> 1);
    for (uint64_t j = 0; j < (n >> 1); j++) {
        uint64_t tx = X[j] + Y[j];
        if (tx >= twoq) tx -= twoq;
        uint64_t ty = X[j] + twoq - Y[j];
        uint64_t a = lazy_mul(tx, inv_n, inv_n_p, q);
        uint64_t b = lazy_mul(ty, inv_n_w, inv_n_w_p, q);
        X[j] = a >= q ? a - q : a;                                 /* :413-432 */
        Y[j] = b >= q ? b - q : b;
    }
}

/* ------------------------------------------------------------------ K3 dyadic multiply */

/* device/mod_ops.hpp:21-29 */
static inline uint64_t ref_addmod(uint64_t a, uint64_t b, uint64_t m) {
    u128 s = (u128)a + b;
    if (s >= m) s -= m;
    return (uint64_t)s;
}

/* device/mod_ops.hpp:31-84 transliterated (len/barr_lo from host/src/fpga.cpp:366-373) */
static inline uint64_t ref_multmod(uint64_t a, uint64_t b, uint64_t m, uint64_t len, uint64_t barr_lo) {
    const uint64_t twice_m = m << 1;
    uint64_t x = a, y = b;
    if (x >= twice_m) x -= twice_m;
    if (x >= m) x -= m;
    if (y >= twice_m) y -= twice_m;
    if (y >= m) y -= m;
    u128 p = (u128)x * y;
    uint64_t lo = (uint64_t)p, hi = (uint64_t)(p >> 64);
    uint64_t c1 = (lo >> len) + (hi << (64 - len));
    uint64_t c3 = mulhi64(c1, barr_lo);
    uint64_t c4 = lo - c3 * m;
    return c4 < m ? c4 : c4 - m;
}

void orc_dyadic_multiply(uint64_t* out, const uint64_t* a, const uint64_t* b, uint64_t n,
                         const uint64_t* moduli, uint64_t n_moduli, int exact) {
    for (uint64_t m = 0; m < n_moduli; m++) {
        const uint64_t q = moduli[m];
        uint64_t len = 0, barr_lo = 0;
        if (!exact) {
            unsigned fl = 63; while (!(q >> fl)) fl--;             /* floor(log2 q) */
            len = fl - 1;
            barr_lo = (uint64_t)(((u128)1 << (len + 64)) / q);
        }
        const uint64_t *x0 = a + m * n, *x1 = a + (n_moduli + m) * n;
        const uint64_t *y0 = b + m * n, *y1 = b + (n_moduli + m) * n;
        uint64_t *r0 = out + m * n, *r1 = out + (n_moduli + m) * n, *r2 = out + (2 * n_moduli + m) * n;
        for (uint64_t j = 0; j < n; j++) {
            if (exact) {                                           /* test_dyadic_multiply.cpp:59-82 */
                uint64_t a0 = x0[j] % q, a1 = x1[j] % q, b0 = y0[j] % q, b1 = y1[j] % q;
                r0[j] = orc_mulmod(a0, b0, q);
                r1[j] = (uint64_t)(((u128)orc_mulmod(a0, b1, q) + orc_mulmod(a1, b0, q)) % q);
                r2[j] = orc_mulmod(a1, b1, q);
            } else {                                               /* dyadic_multiply.cpp:195-228 */
                r0[j] = ref_multmod(x0[j], y0[j], q, len, barr_lo);
                r1[j] = ref_addmod(ref_multmod(x0[j], y1[j], q, len, barr_lo),
                                   ref_multmod(x1[j], y0[j], q, len, barr_lo), q);
                r2[j] = ref_multmod(x1[j], y1[j], q, len, barr_lo);
            }
        }
    }
}

/* ------------------------------------------------------------------ K4 keyswitch */

/* device/mod_ops.hpp:206-224 */
static inline uint64_t addmod(uint64_t x, uint64_t y, uint64_t q) { uint64_t s = x + y; return s >= q ? s - q : s; }
static inline uint64_t submod(uint64_t x, uint64_t y, uint64_t q) { uint64_t d = x + q - y; return d >= q ? d - q : d; }
/* device/mod_ops.hpp:213-217 with q_barr = floor(2^64/q) (fpga.cpp:1053) */
static inline uint64_t barrett64(uint64_t v, uint64_t q, uint64_t q_barr) {
    uint64_t r = v - mulhi64(v, q_barr) * q;
    return r >= q ? r - q : r;
}
/* device/mod_ops.hpp:226-269, factor 8 */
static inline uint64_t reduce8(uint64_t x, uint64_t q) {
    if (x >= 4 * q) x -= 4 * q;
    if (x >= 2 * q) x -= 2 * q;
    if (x >= q) x -= q;
    return x;
}

/* exact CT forward NTT, twiddle index m+i: device/keyswitch/ntt_core.hpp:222,285-291 */
void orc_ks_ntt(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* roots) {
    uint64_t t = n >> 1;
    for (uint64_t m = 1; m < n; m <<= 1, t >>= 1)
        for (uint64_t i = 0; i < m; i++) {
            const uint64_t W = roots[m + i];
            uint64_t *X = x + 2 * i * t, *Y = X + t;
            for (uint64_t j = 0; j < t; j++) {
                uint64_t wy = orc_mulmod(Y[j], W, q);
                uint64_t a = addmod(X[j], wy, q), b = submod(X[j], wy, q);
                X[j] = a; Y[j] = b;
            }
        }
}

/* exact GS inverse NTT, all log2(n) stages with the inverse table read from index 0
 * (device/keyswitch/intt_core.hpp:118,274,335-347,436), then * n^-1 (:72-93) */
void orc_ks_intt(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* inv0) {
    uint64_t t = 1, acc = 0;
    for (uint64_t m = n >> 1; m >= 1; m >>= 1, t <<= 1) {
        for (uint64_t i = 0; i < m; i++) {
            const uint64_t W = inv0[acc + i];
            uint64_t *X = x + 2 * i * t, *Y = X + t;
            for (uint64_t j = 0; j < t; j++) {
                uint64_t a = addmod(X[j], Y[j], q);
                uint64_t b = orc_mulmod(submod(X[j], Y[j], q), W, q);
                X[j] = a; Y[j] = b;
            }
        }
        acc += m;
    }
    const uint64_t inv_n = orc_invmod(n, q);                      /* fpga.cpp:1070-1089 */
    for (uint64_t j = 0; j < n; j++) x[j] = orc_mulmod(x[j], inv_n, q);
}

int orc_keyswitch(uint64_t* result, const uint64_t* t_target, uint64_t n, uint64_t L, uint64_t K,
                  uint64_t rns, uint64_t kcc, const uint64_t* moduli, const uint64_t* const* keys,
                  const uint64_t* modswitch, const uint64_t* twiddles) {
    if (kcc != 2 || L == 0 || K < 2 || L >= K || rns == 0) return -1;   /* keyswitch.cpp:21-34 */
    const uint64_t sp = K - 1;                   /* special prime slot: load.hpp:79-84 */
    const uint64_t q_sp = moduli[sp];
    uint64_t* tw_own = NULL;
    if (!twiddles) {                              /* fpga.cpp:1097-1109 */
        tw_own = (uint64_t*)malloc(K * 4 * n * sizeof(uint64_t));
        for (uint64_t i = 0; i < K; i++)
            orc_tables_keyswitch(n, moduli[i], orc_minimal_primitive_root(2 * n, moduli[i]), tw_own + i * 4 * n);
        twiddles = tw_own;
    }
    uint64_t* c = (uint64_t*)malloc(L * n * sizeof(uint64_t));           /* step 1 outputs */
    uint64_t* u = (uint64_t*)malloc(n * sizeof(uint64_t));
    uint64_t* prod = (uint64_t*)calloc(2 * (L + 1) * n, sizeof(uint64_t)); /* [k][slot][j] */
    uint64_t* s = (uint64_t*)malloc(n * sizeof(uint64_t));

    /* 1. c_d = INTT_{q_d}(t_target[d]) */
    for (uint64_t d = 0; d < L; d++) {
        memcpy(c + d * n, t_target + d * n, n * sizeof(uint64_t));
        orc_ks_intt(c + d * n, n, moduli[d], twiddles + d * 4 * n);
    }
    /* 2+3. per RNS slot: NTT of the mod-reduced c_d, multiply-accumulate with the key */
    for (uint64_t slot = 0; slot <= L; slot++) {
        const uint64_t i = slot < L ? slot : sp;
        const uint64_t q = moduli[i], qb = orc_shoup_factor(1, q);
        for (uint64_t d = 0; d < L; d++) {
            for (uint64_t j = 0; j < n; j++) u[j] = barrett64(c[d * n + j], q, qb);   /* intt1_redu.hpp:36-42 */
            orc_ks_ntt(u, n, q, twiddles + i * 4 * n + 2 * n);
            for (uint64_t k = 0; k < 2; k++) {                                          /* dyadmult.hpp:128-140 */
                const uint64_t* key = keys[d] + (k * K + i) * n;                        /* fpga.cpp:1186-1190 */
                uint64_t* p = prod + (k * (L + 1) + slot) * n;
                for (uint64_t j = 0; j < n; j++) p[j] = addmod(orc_mulmod(u[j], key[j], q), p[j], q);
            }
        }
    }
    /* 4-7 per key component */
    const uint64_t half = q_sp >> 1;
    for (uint64_t k = 0; k < 2; k++) {
        memcpy(s, prod + (k * (L + 1) + L) * n, n * sizeof(uint64_t));
        orc_ks_intt(s, n, q_sp, twiddles + sp * 4 * n);
        for (uint64_t j = 0; j < n; j++) s[j] = addmod(s[j], half, q_sp);               /* intt2_redu.hpp:25,43 */
        for (uint64_t i = 0; i < L; i++) {
            const uint64_t q = moduli[i], qb = orc_shoup_factor(1, q);
            const uint64_t fix = q - barrett64(half, q, qb);                            /* intt2_redu.hpp:31-32 */
            for (uint64_t j = 0; j < n; j++) u[j] = barrett64(s[j] + fix, q, qb);       /* :49-51 */
            orc_ks_ntt(u, n, q, twiddles + i * 4 * n + 2 * n);
            const uint64_t msf = reduce8(modswitch[i], q);                              /* fpga.cpp:1057-1061 */
            const uint64_t* p = prod + (k * (L + 1) + i) * n;
            uint64_t* res = result + (k * L + i) * n;                                   /* fpga.cpp:452-468 */
            for (uint64_t j = 0; j < n; j++) {
                uint64_t in = reduce8(p[j] + 4 * q - u[j], q);                          /* ms.hpp:70-82 */
                uint64_t out = orc_mulmod(in, msf, q);
                uint64_t r = res[j] + out;                                              /* fpga.cpp:453-457 */
                res[j] = r >= q ? r - q : r;
            }
        }
    }
    free(c); free(u); free(prod); free(s); free(tw_own);
    return 0;
}

/* ------------------------------------------------------------------ helpers */

uint64_t orc_fnv1a64(const void* data, size_t nbytes) {
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

void orc_fill_splitmix(uint64_t* x, size_t count, uint64_t seed, uint64_t q) {
    uint64_t s = seed;
    for (size_t i = 0; i < count; i++) {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        x[i] = q ? z % q : z;
    }
}
