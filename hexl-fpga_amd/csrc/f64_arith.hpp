// f64_arith.hpp -- exact modular arithmetic on 64-bit FLOATING-POINT lanes for moduli p < 2^52.
//
// Why: the keyswitch is ALU-bound on MI355X, and a 64-bit integer Harvey butterfly costs ~36 VALU
// instructions there (a 64x64 multiply is 4-5 v_mad_u64_u32/v_mul_lo_u32 plus carry chains), while
// v_fma_f64 / v_mul_f64 / v_add_f64 / v_rndne_f64 each handle a whole 64-bit lane value in one
// instruction at about the same issue cost. The reference restricts keyswitch moduli to <= 2^52
// (host/src/keyswitch.cpp:32), exactly the range where a double holds every residue exactly, so the
// whole pipeline can run on the FP64 pipe with EXACT integer results:
//   * every value is an integer of magnitude < 2^53 stored in a double (residues are kept
//     "centred", |x| <= ~p/2, between operations);
//   * a product a*b is split error-free by FMA: h = fl(a*b), l = fma(a,b,-h), a*b = h + l exactly;
//   * the quotient k ~ a*b/p comes from one rounded multiply by a precomputed reciprocal and
//     v_rndne; r = fma(-k,p,h) + l is then exact because |h - k*p| < 2^53 (bounds below).
// Results are the mathematical residues, hence bit-identical to the reference's canonical
// AddUIntMod/SubUIntMod/MultiplyUIntMod dataflow (device/keyswitch/*.hpp) after the final lift to
// [0,p). The standalone _NTT/_INTT entry points keep the integer Harvey kernels: their contract
// includes uint64 wrap-around behaviour on arbitrary tables, which only integer code reproduces.
//
// The same source compiles for the host (g++ -mfma, used by tests/cpp/f64_selftest.cpp to validate
// the bounds against exact integer arithmetic on adversarial inputs) and for gfx950.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HX_HD __host__ __device__ __forceinline__
#else
#define HX_HD static inline __attribute__((always_inline))
#endif

namespace hxf {

struct Mod {        // per-modulus constants
    double p;       // modulus (exact, p < 2^52)
    double pinv;    // fl(1/p)
};
struct InvScale {   // inverse-transform scaling: n^-1 and n^-1 * W_last (centred) with their fl(./p) factors
    double n, n_p, nw, nw_p;
};

// x -> x - p*rint(x/p): |result| <= p/2 + 2 for any integer |x| < 2^53. Exact: the result is an
// integer of magnitude < 2^53, so the fma's single rounding is the identity.
HX_HD double reduce(double x, const Mod m) { return __builtin_fma(-__builtin_rint(x * m.pinv), m.p, x); }

// centred (|x| <= p/2 + 2) -> canonical [0, p). floor(x/p) is -1 exactly when x < 0: the rounded
// quotient of a non-zero x keeps its sign and |x/p| < 1.
HX_HD double lift(double x, const Mod m) { return __builtin_fma(-__builtin_floor(x * m.pinv), m.p, x); }

// x*w mod p for a constant w with precomputed wp = fl(w/p) ("Shoup" form). Requires |w| <= p/2 and
// |x| <= 1.5p. |result| <= (0.5 + |x|/(2p)) * p.
//   exactness: |h - k*p| <= |result| + |l|, |l| <= ulp(h)/2 <= 2^50, |result| <= 1.25p  =>  < 2^53.
HX_HD double mul_shoup(double x, double w, double wp, const Mod m) {
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    const double k = __builtin_rint(x * wp);
    return __builtin_fma(-k, m.p, h) + l;
}

// a*b mod p for two variables, |a|,|b| <= p/2 + 2: |result| <= 0.7p.
HX_HD double mul_mod(double a, double b, const Mod m) {
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    const double k = __builtin_rint(h * m.pinv);
    return __builtin_fma(-k, m.p, h) + l;
}

// uint64 <-> double. to_f64 is exact for x < 2^53 (all in-domain data); from_f64 needs 0 <= x < 2^52.
HX_HD double to_f64(uint64_t x) { return __builtin_fma((double)(uint32_t)(x >> 32), 4294967296.0, (double)(uint32_t)x); }
HX_HD uint64_t from_f64(double x) {
    const double t = x + 4503599627370496.0;                 // 2^52: the integer lands in the mantissa
    uint64_t b;
    __builtin_memcpy(&b, &t, 8);
    return b & 0x000FFFFFFFFFFFFFull;
}

// ---- butterflies (both outputs centred) ---------------------------------------------------------
// Cooley-Tukey / forward (device/keyswitch/ntt_core.hpp:285-291):  X' = X + W*Y, Y' = X - W*Y
HX_HD void ct_bfly(double& X, double& Y, double w, double wp, const Mod m) {
    const double t = mul_shoup(Y, w, wp, m);                 // |t| <= 0.75p
    const double a = X + t, b = X - t;                       // |.| <= 1.25p + 2
    X = reduce(a, m);
    Y = reduce(b, m);
}
// Gentleman-Sande / inverse (device/keyswitch/intt_core.hpp:335-347):  X' = X + Y, Y' = (X - Y)*W
HX_HD void gs_bfly(double& X, double& Y, double w, double wp, const Mod m) {
    const double s = X + Y, d = X - Y;                       // |.| <= p + 4
    X = reduce(s, m);
    Y = reduce(mul_shoup(d, w, wp, m), m);                   // |product| <= p before the reduce
}

}  // namespace hxf
