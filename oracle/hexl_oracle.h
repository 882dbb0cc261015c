/*
 * hexl_oracle.h -- CPU restatement of the hexl-fpga hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X kernels. It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The shipped library (libhexl_mi355x.so / libhexl-fpga.so) never links or calls it.
 *
 * Every function cites the reference file:line (relative to the intel/hexl-fpga v2.0
 * tree) whose behaviour it restates.
 *
 * Parity pins:
 *   - fwd/inv NTT + tables : pinned against the reference's own CPU oracle
 *     (tests/test_utils/ntt.cpp, built into oracle/_ref by oracle/Makefile) and the
 *     FNV-1a digests in tests/golden/ntt_digests.json captured from it.
 *   - dyadic multiply      : pinned against the reference test's inline `%` model
 *     (tests/test_dyadic_multiply.cpp:59-82) and a transliteration of MultMod.
 *   - keyswitch            : PARITY UNPINNED against reference vectors (testdata.zip is
 *     an external download, README.md:166-176, absent offline). Pinned instead by an
 *     independent schoolbook model and an RLWE decrypt check in tests/.
 */
#ifndef HEXL_ORACLE_H
#define HEXL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- number theory (host/src/number_theory_util.cpp, tests/test_utils/ntt.cpp:14-243) ---- */
uint64_t orc_mulmod(uint64_t x, uint64_t y, uint64_t q);
uint64_t orc_powmod(uint64_t base, uint64_t exp, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q);
int      orc_is_prime(uint64_t n);
/* primes in (2^bits, 2^(bits+1)) with p = 1 mod 2*ntt_size; returns count found */
size_t   orc_generate_primes(uint64_t* out, size_t num, unsigned bits, uint64_t ntt_size);
uint64_t orc_minimal_primitive_root(uint64_t degree, uint64_t q);
uint64_t orc_reverse_bits(uint64_t x, unsigned width);
/* floor(op * 2^64 / q)  (MultiplyFactor(op, 64, q).BarrettFactor()) */
uint64_t orc_shoup_factor(uint64_t op, uint64_t q);

/* ---- twiddle tables ---- */
/* HEXL layout used by the standalone _NTT/_INTT callers (tests/test_utils/ntt.cpp:290-384):
 * inverse table indexed from 1, inv[0]=1, inv[n-1]=W^-1 of the last stage. */
void orc_tables_hexl(uint64_t n, uint64_t q, uint64_t w, uint64_t* roots, uint64_t* precon,
                     uint64_t* inv_roots, uint64_t* inv_precon);
/* hexl-fpga keyswitch layout (host/src/twiddle-factors.cpp:16-62): block of 4n words
 * [inv_roots | precon64_inv | roots | precon64_roots]; inverse indexed from 0, inv[n-1]=0,
 * precon64_roots[0]=0. */
void orc_tables_keyswitch(uint64_t n, uint64_t q, uint64_t w, uint64_t* block4n);

/* ---- K1/K2: Harvey lazy NTT, op-for-op uint64 wrap-around semantics ---- */
/* device/fwd_ntt.cpp:137-493 == tests/test_utils/ntt.cpp:474-548 (output in [0,q) for
 * in-range inputs; deterministic garbage for out-of-range inputs, compared exactly). */
void orc_ntt_fwd(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* roots,
                 const uint64_t* precon);
/* device/inv_ntt.cpp:141-441 == tests/test_utils/ntt.cpp:580-659; inv_n, inv_n_w are
 * caller-supplied scalars as in host/inc/hexl-fpga.h:150-154. */
void orc_ntt_inv(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* inv_roots,
                 const uint64_t* inv_precon, uint64_t inv_n, uint64_t inv_n_w);

/* ---- K3: dyadic multiply (device/dyadic_multiply.cpp:195-228, mod_ops.hpp:21-84) ---- */
/* layout: op[(p*n_moduli + m)*n + j], p<2; out[(p*n_moduli + m)*n + j], p<3.
 * exact=1: mathematically exact (a mod q)(b mod q) mod q for any operands;
 * exact=0: transliteration of the reference MultMod/AddMod with host len/barr_lo
 *          (host/src/fpga.cpp:366-373) -- defined only for operands < 4q. */
void orc_dyadic_multiply(uint64_t* out, const uint64_t* a, const uint64_t* b, uint64_t n,
                         const uint64_t* moduli, uint64_t n_moduli, int exact);

/* ---- K4: keyswitch (device/keyswitch/ headers, fpga.cpp:441-475,
 *      1049-1123) ----
 * t_target[d*n+j], d<L; result[(k*L+i)*n+j], k<2, accumulated into (+= mod q_i);
 * keys[d][(k*K+i)*n+j]; moduli[K], modswitch[K]; twiddles = K blocks of 4n words in the
 * hexl-fpga layout or NULL (derived from MinimalPrimitiveRoot). Returns 0 on success. */
int orc_keyswitch(uint64_t* result, const uint64_t* t_target, uint64_t n, uint64_t L,
                  uint64_t K, uint64_t rns, uint64_t kcc, const uint64_t* moduli,
                  const uint64_t* const* keys, const uint64_t* modswitch,
                  const uint64_t* twiddles);

/* canonical (exact) negacyclic NTT/INTT used inside the keyswitch (ntt_core.hpp:285-291,
 * intt_core.hpp:335-347 + :72-93). Tables are blocks 2 and 0 of the keyswitch layout. */
void orc_ks_ntt(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* roots);
void orc_ks_intt(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* inv_roots_from0);

/* ---- helpers shared by tests/bench ---- */
uint64_t orc_fnv1a64(const void* data, size_t nbytes);
/* splitmix64 stream: x[i] = next(state) % q (q==0 -> raw 64-bit values) */
void orc_fill_splitmix(uint64_t* x, size_t count, uint64_t seed, uint64_t q);

#ifdef __cplusplus
}
#endif
#endif
