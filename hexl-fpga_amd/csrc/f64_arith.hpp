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
#include <hip/hip_runtime.h>
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

// a*b mod p for two variables, |a|,|b| <= p/2 + 2: |result| <= 0.7p (general bound in the LAZY notes below).
HX_HD double mul_mod(double a, double b, const Mod m) {
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    const double k = __builtin_rint(h * m.pinv);
    return __builtin_fma(-k, m.p, h) + l;
}

// uint64 <-> double. to_f64 is exact for x < 2^53 (all in-domain data); from_f64 needs 0 <= x < 2^52.
HX_HD double to_f64(uint64_t x) { return __builtin_fma((double)(uint32_t)(x >> 32), 4294967296.0, (double)(uint32_t)x); }
// the same for words KNOWN to be below 2^52 (every in-range residue: moduli < 2^52): the word becomes the mantissa of
// 2^52 + x (one 32-bit OR on the high half), one FP64 subtraction takes the 2^52 off again -- two instructions, one of them
// at the 32-bit rate, instead of two conversions and an fma. A word >= 2^52 gives garbage (callers flag it, to_f64_checked).
HX_HD double to_f64_lt52(uint64_t x) {
    const uint64_t b = x | 0x4330000000000000ull;
    double t;
    __builtin_memcpy(&t, &b, 8);
    return t - 4503599627370496.0;
}
HX_HD uint64_t from_f64(double x) {
    const double t = x + 4503599627370496.0;                 // 2^52: the integer lands in the mantissa
    uint64_t b;
    __builtin_memcpy(&b, &t, 8);
    return b & 0x000FFFFFFFFFFFFFull;
}

// The STRICT butterflies (every value reduced after every operation) leave room above 2^52: the largest intermediate is the
// inverse butterfly's |h - k p| <= (1.31 + 0.22) p for |d| = |X - Y| <= p + 4 (quotient from the product: three roundings of a
// value near p/2, ulp 1/2), the forward's |X + t| <= 1.4 p, all below 2^53 up to p ~ 2^52.39. The standalone _NTT / _INTT fast
// path (ntt.hip) takes moduli up to 2^52 * 1.125 on them -- SURVEY 8d's q = 2^52 + 393217 among them -- which only needs word
// <-> double conversions that do not assume 52 bits. tests/cpp/f64_selftest.cpp replays both transforms at the bound against
// the oracle and tracks the largest magnitude.
constexpr uint64_t STRICT_NTT_MAX_Q = 5066549580791808ull;      // 2^52 * 1.125
// an integer 0 <= x < 2^53 held in a double -> uint64 (from_f64 needs x < 2^52)
HX_HD uint64_t from_f64_53(double x) {
    const double hi = __builtin_floor(x * (1.0 / 4294967296.0));
    const double lo = __builtin_fma(-hi, 4294967296.0, x);
    return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}

// The precondition of the FP64 kernels -- every input word below its modulus -- checked where the word is converted anyway:
// one compare per word (NaN-safe form; words >= 2^53 convert inexactly but stay >= p), collected as a wave-uniform lane mask
// (compare into a scalar register pair + scalar OR: no vector registers). The kernels OR the outcome into a per-plan flag
// that hexl_ks_range_check() reads; results for out-of-range words are outside the contract either way.
#if defined(__HIPCC__)
typedef unsigned long long RangeMask;
__device__ __forceinline__ double to_f64_checked(uint64_t x, const Mod m, RangeMask& out_of_range) {
    const double d = to_f64(x);
    out_of_range |= __builtin_amdgcn_ballot_w64(!(d < m.p));
    return d;
}
// the two-instruction conversion with the range check on the integer word (q = the modulus as an integer, < 2^52)
__device__ __forceinline__ double to_f64_lt52_checked(uint64_t x, uint64_t q, RangeMask& out_of_range) {
    out_of_range |= __builtin_amdgcn_ballot_w64(x >= q);
    return to_f64_lt52(x);
}
__device__ __forceinline__ void report_range(RangeMask out_of_range, unsigned* flag) {
    if (out_of_range != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
#endif

// ---- butterflies (both outputs centred) ---------------------------------------------------------
// Cooley-Tukey / forward (device/keyswitch/ntt_core.hpp:285-291):  X' = X + W*Y, Y' = X - W*Y.
// The forward butterflies take the quotient from the product itself (mul_mod) instead of a second table of
// w/p: same six instructions, half the twiddle traffic and registers. (The inverse keeps the Shoup form: its
// bound chain below needs the smaller error term.)
HX_HD void ct_bfly(double& X, double& Y, double w, const Mod m) {
    const double t = mul_mod(Y, w, m);                       // |t| <= 0.7p
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

// "Semi-strict" forward butterfly for moduli ABOVE the lazy bound (up to 2^52 (1 + 2^-20), SEMI_MAX_MODULUS): the product takes
// the Shoup form (w/p table: two roundings instead of three, |y w mod p| <= (0.5 + 0.504 |y|/p) p at p = 2^52), and the two outputs
// are range-reduced only when the NEXT stage uses them as the ADDED operand (both outputs of a butterfly have the same role in the
// next stage); outputs that the next stage multiplies stay as they are. Within a register pass that starts from reduced values
// (|x| <= p/2 + 2) the multiplied operand grows as y -> 1 + 0.504 y: 0.5, 1.252, 1.631, 1.822, and the pass's last stage reduces
// everything (|X + t| <= 0.5 + 0.5 + 0.504 * 1.822 = 1.918 p < 2^53 = 2 p). Exactness of the product: |h - k p| <= (0.5 + 0.504 *
// 1.822) p + |l| (<= 2^50) = 1.67 p. 11 instead of 14 FP64 instructions per butterfly on average over a four-stage pass.
// tests/cpp/f64_selftest.cpp replays whole transforms on this schedule at the largest 52-bit primes and at 2^52 + 393217.
constexpr double SEMI_MAX_MODULUS = 4503599627370496.0 * (1.0 + 1.0 / 1048576.0);
HX_HD void ct_bfly_semi(double& X, double& Y, double w, double wp, const Mod m, bool reduce_outputs) {
    const double t = mul_shoup(Y, w, wp, m);
    const double a = X + t, b = X - t;
    X = reduce_outputs ? reduce(a, m) : a;
    Y = reduce_outputs ? reduce(b, m) : b;
}

// ---- lazy variants for moduli p <= 2^51 * (1 + 2^-7)  ("LAZY" regime) ------------------------------------
// Every value only has to stay an exactly representable integer, |x| < 2^53 ~ 3.97p here, which leaves room
// to skip most range reductions:
//   mul_shoup bound in this regime: |x*w mod p| <= (0.5 + 0.252*|x|/p) * p   (two roundings of |x*w/p| <= |x|/2)
//   mul_mod bound (|w| <= p/2):     |x*w mod p| <= (0.5 + 0.378*|x|/p) * p   (the quotient is taken from h = fl(x*w),
//             so the split-off low part |l| <= ulp(h)/2 <= 0.126*|x| adds to the remainder); exact because
//             |h - k*p| <= (0.5 + 0.252*|x|/p) * p < 2^53
//   forward:  butterflies (mul_mod) without reduce grow the bound c (|x| <= c*p) as c -> 1.378c + 0.5:
//             0.5 -> 1.189 -> 2.139 -> 3.447 (< 3.97); a reduce of every element after each third stage
//             (and after the last) restarts the chain. 8 FP64 ops per butterfly + 6 every third stage.
//             Without the reduce after the last stage the outputs are bounded by 2.14p.
//   inverse:  sums double, so the sum output is reduced every stage (-> 0.5p) and the product output never:
//             inputs <= p  =>  |X+Y|,|X-Y| <= 2p  =>  product <= (0.5 + 0.504)p ~ p: stable at c = 1. 11 ops.
//   exactness of mul_shoup: |h - k*p| <= |result| + |l| <= 1.01p + 2^49 < 2^53.
// tests/cpp/f64_selftest.cpp replays both schedules against exact integers and records the largest |x| seen.
constexpr double LAZY_MAX_MODULUS = 2251799813685248.0 * (1.0 + 1.0 / 128.0);   // 2^51 * (1 + 2^-7)

HX_HD void ct_bfly_lazy(double& X, double& Y, double w, const Mod m) {
    const double t = mul_mod(Y, w, m);
    const double a = X + t, b = X - t;
    X = a;
    Y = b;
}
HX_HD void gs_bfly_lazy(double& X, double& Y, double w, double wp, const Mod m) {
    const double s = X + Y, d = X - Y;
    X = reduce(s, m);
    Y = mul_shoup(d, w, wp, m);
}
// Inverse butterflies WITHOUT the w/p table (round 4): the quotient comes from the product itself, as in the forward
// transforms -- same instruction count, half the per-lane twiddle bytes and registers of an inverse pass.
//   bound chain, top tier (a = p 2^-53 = 0.252): a product output is <= (0.5 + 0.378 |d|/p) p with |d| = |X - Y| <= 2 y when
//   both inputs are product outputs of the previous stage (<= y p):  y -> 0.5 + 0.756 y = 0.5, 0.878, 1.164, 1.380, 1.543, 1.667,
//   1.760, 1.831, ... -> 2.05: |d| would pass 2^53 = 3.97p after twelve stages. One stage (INV_NOWP_STRICT_STAGE, the 6th
//   global stage of every transform that has more) reduces its product outputs as well, which restarts the chain: at most
//   eight lazy stages follow (n <= 2^15), y <= 1.884, |d| <= 3.77p. Lower tiers (a <= 0.125): y -> 0.5 + 0.375 y <= 0.8.
//   exactness of mul_mod: |h - k p| <= (0.5 + 0.252 |d|/p) p <= 1.5p. The last stage (n^-1 folded in) keeps mul_shoup on
//   scalar constants. tests/cpp/f64_selftest.cpp replays the schedule against the oracle and tracks the largest |x|.
constexpr int INV_NOWP_STRICT_STAGE = 6;
HX_HD void gs_bfly_lazy_nowp(double& X, double& Y, double w, const Mod m) {
    const double s = X + Y, d = X - Y;
    X = reduce(s, m);
    Y = mul_mod(d, w, m);
}
HX_HD void gs_bfly_nowp(double& X, double& Y, double w, const Mod m) {     // both outputs centred (strict kernels; the strict stage)
    const double s = X + Y, d = X - Y;
    X = reduce(s, m);
    Y = reduce(mul_mod(d, w, m), m);
}
// Folded multiply-accumulate (LAZY regime; the key multiply-accumulate of the slot-major keyswitch): acc + x*k mod p in
// EIGHT operations instead of mul_mod + add + reduce = ten -- the accumulator enters the quotient estimate, so the result
// is reduced in the same step:  acc' = (h - K p) + (acc + l),  K = rint(fl(h/p) + acc/p),  h + l = x*k exactly.
// Requires |k| <= p/2, |x| <= c p with c as in the forward schedule above (un-reduced transform output), |acc| <= 1.6p.
//   K differs from (x k + acc)/p by at most 0.5 + (three roundings of a value < 2^52: 0.75) + |l|/p  =>  |acc'| <= 1.6p
//   exact: acc + l is an integer below 1.6p + 2^49 < 2^52; |h - K p| <= |acc'| + |acc + l| < 3.4p < 2^53; acc' an integer < 2^53.
// tests/cpp/f64_selftest.cpp replays chains of it against 128-bit integers at extreme operands for every tier.
// STRICT tier (moduli up to 2^52, transform output reduced: |x| <= p/2 + 2, |k| <= p/2): the quotient estimate is off by at most
// 0.5 + three roundings of a value below 2^50 (ulp 1/4: 0.375)  =>  |acc'| <= 0.875p + |l| whatever the accumulator was; exact:
// |acc + l| <= 0.9p + 2^48 < 2^53, |h - K p| <= |acc'| + |acc + l| <= 1.8p < 2^53. The consumer of the last term reduces once.
HX_HD double mac_fold(double acc, double x, double k, const Mod m) {
    const double h = x * k;
    const double l = __builtin_fma(x, k, -h);
    const double s = acc + l;
    const double K = __builtin_rint(__builtin_fma(acc, m.pinv, h * m.pinv));
    return __builtin_fma(-K, m.p, h) + s;
}

// forward schedule: reduce every element after global stage s (1-based) when (s + shift) % period == 0 or s is the last stage.
// shift = 1 (round 4, the mod-up transforms of the slot-major keyswitch) is the schedule for inputs that are NOT centred:
// canonical residues of a neighbouring modulus, |x| <= rho p with rho = max q / min q <= LAZY_SKIP_MAX_RATIO, taken as they
// are (no range reduction of the input). The first group is one stage shorter, every later one is a full period:
//   top tier (growth c -> 1.378 c + 0.5, limit 3.97): 1.25 -> 2.22 -> 3.56 | 0.5 -> ... -> 3.45 | ...; an un-reduced tail is at
//   most `period` stages (3.45p; 6.22p / 11.8p in the lower tiers), which mac_fold accepts (its bounds only need
//   |x k| < 2^103 and |x k / p| < 2^52).  period 6 (1.1875 c + 0.5, limit 8): 1.25 -> 6.58 after five stages; period 12
//   (1.09375 c + 0.5, limit 16): 1.25 -> 12.3 after eleven.
HX_HD constexpr bool lazy_fwd_reduce_after(int s, int logn, int period = 3, int shift = 0) {
    return ((s + shift) % period == 0) || (s == logn);
}
constexpr double LAZY_SKIP_MAX_RATIO = 1.25;

// Smaller moduli leave more head room: with a = p * 2^-53 a forward butterfly grows the bound as
// c -> (1 + 1.5a) c + 0.5 and every value must stay below c = 1/a (|x| < 2^53):
//   p <= 2^51(1+2^-7): a = 0.252, limit 3.97: 0.5 -> 1.19 -> 2.14 -> 3.45                         period 3
//   p <= 2^50:         a = 0.125, limit 8:    0.5 -> 1.09 -> 1.80 -> 2.64 -> 3.63 -> 4.81 -> 6.22  period 6
//   p <= 2^49:         a = 0.0625, limit 16:  ... -> 11.8 after twelve stages                      period 12
// (consumers of an un-reduced tail need c + 0.5 < 1/a: at most two tail stages follow a reduction for n <= 2^14.)
// The PERIODIC inverse schedule does not change with the tier (the I schedules further down do). tests/cpp/f64_selftest.cpp replays every tier.
HX_HD constexpr int lazy_period_for(double max_modulus) {
    return max_modulus <= 562949953421312.0 ? 12 : max_modulus <= 1125899906842624.0 ? 6 : max_modulus <= LAZY_MAX_MODULUS ? 3 : 0;
}


// ---- the bound chains above, evaluated at compile time -----------------------------------------------------------------
// The lazy schedules are exact only while every intermediate stays below 2^53 = p / a, a = p 2^-53. The margins are thin at the top of
// each tier (3.77p of 3.97p), and they depend on constants that live in different places -- LAZY_MAX_MODULUS, the tier boundaries of
// lazy_period_for, LAZY_SKIP_MAX_RATIO, INV_NOWP_STRICT_STAGE, the accumulator bound of mac_fold -- so the recurrences are replayed here
// with static_asserts: changing a ratio, a tier boundary or a period without re-deriving the bounds fails the build instead of relying on
// the randomised host replay (tests/cpp/f64_selftest.cpp) to notice. `a` is taken at the LARGEST modulus of the tier (the worst case).
namespace bounds {
constexpr double TWO53 = 9007199254740992.0;
constexpr double tier_top(int period) { return period == 12 ? 562949953421312.0 : period == 6 ? 1125899906842624.0 : LAZY_MAX_MODULUS; }
constexpr double tier_a(int period) { return tier_top(period) / TWO53; }
constexpr double SLOP = 1e-9;                                      // the "+ 2" of "|x| <= p/2 + 2" relative to p >= 2^48 ... and to spare
// forward butterflies without a range reduction: |x| <= c p  ->  (1 + 1.5 a) c + 0.5 per stage
constexpr double fwd_chain(double c, double a, int stages) {
    for (int s = 0; s < stages; ++s) c = (1.0 + 1.5 * a) * c + 0.5;
    return c;
}
// inverse butterflies whose quotient comes from the product (gs_bfly_lazy_nowp): product outputs y -> 0.5 + 3 a y, |X - Y| <= 2 y
constexpr double inv_chain(double y, double a, int stages) {
    for (int s = 0; s < stages; ++s) y = 0.5 + 3.0 * a * y;
    return y;
}
constexpr double MAC_FOLD_ACC = 1.7;                               // |acc| handed to the mod-down epilogue by mac_fold (lazy tiers)
constexpr bool tier_ok(int period) {
    const double a = tier_a(period), limit = 1.0 / a;
    // (i) standard schedule: the sums of `period` stages from a centred value (the last of them is the one that gets reduced)
    if (!(fwd_chain(0.5 + SLOP, a, period) < limit)) return false;
    // (ii) shifted schedule (SKIP mod-up): a canonical residue of a neighbouring modulus, rho p; the first group is period - 1 stages
    if (!(fwd_chain(LAZY_SKIP_MAX_RATIO, a, period - 1) < limit)) return false;
    // (iii) SKIP mod-down: the centred special-prime remainder, 0.5 rho p, on the standard schedule for `period` stages
    if (!(fwd_chain(0.5 * LAZY_SKIP_MAX_RATIO + SLOP, a, period) < limit)) return false;
    // (iv) mod-down epilogue: un-reduced accumulator minus an un-reduced transform tail of at most period - 1 stages
    if (!(MAC_FOLD_ACC + fwd_chain(0.5 + SLOP, a, period - 1) < limit)) return false;
    return true;
}
static_assert(tier_ok(3), "period 3 tier (p <= LAZY_MAX_MODULUS): a forward bound chain passes 2^53");
static_assert(tier_ok(6), "period 6 tier (p <= 2^50): a forward bound chain passes 2^53");
static_assert(tier_ok(12), "period 12 tier (p <= 2^49): a forward bound chain passes 2^53");
static_assert(lazy_period_for(tier_top(3)) == 3 && lazy_period_for(tier_top(6)) == 6 && lazy_period_for(tier_top(12)) == 12 &&
              lazy_period_for(tier_top(3) + 1.0) == 0, "tier boundaries of lazy_period_for moved: re-derive tier_top");
// inverse transforms without the w/p table: INV_NOWP_STRICT_STAGE - 1 lazy stages in front of the strict stage, at most
// 15 - 1 - INV_NOWP_STRICT_STAGE behind it (n <= 2^15; the last stage is the fused scaling on reduced sums), each with 2 y < 2^53 / p
static_assert(2.0 * inv_chain(0.5 + SLOP, tier_a(3), INV_NOWP_STRICT_STAGE - 1) < 1.0 / tier_a(3), "inverse chain in front of the strict stage");
static_assert(2.0 * inv_chain(0.5 + SLOP, tier_a(3), 15 - 1 - INV_NOWP_STRICT_STAGE) < 1.0 / tier_a(3), "inverse chain behind the strict stage");
// canonical inputs taken as they are (rho p: the standalone _INTT fast path and k_ksx_intt with rho = 1): the first stage's product input
static_assert(LAZY_SKIP_MAX_RATIO < 1.0 / tier_a(3) && 2.0 * inv_chain(LAZY_SKIP_MAX_RATIO, tier_a(3), INV_NOWP_STRICT_STAGE - 1) < 1.0 / tier_a(3),
              "inverse chain from un-centred inputs");
}  // namespace bounds

// ---- "X schedules" (round 6): range-reduce only the ADDED operand, only where the chain needs it -------------------------------------
// In a forward butterfly X' = X + t, Y' = X - t, t = mul_mod(Y, w), the multiplied operand never needs a range reduction: |t| <= (0.5 +
// 1.5 a |Y| / p) p for any |Y| < 2^53 (a = p 2^-53). Only X carries the bound forward. The periodic schedules above reduce BOTH outputs
// of every butterfly after each period-th stage (6 instructions per butterfly); an X schedule instead reduces, in front of chosen stages,
// just the X input (3 instructions):
//     N  nothing                          |x| <= c p  ->  (1 + 1.5 a) c + 0.5
//     X  X = reduce(X) before the stage               ->  (0.5 + 0.5) + 1.5 a c
//     F  X and Y reduced before the stage             ->  (0.5 + 0.5) + 0.75 a
// and every value must stay below 2^53 = p / a. tools/gen_xsched.py brute-forces the cheapest schedule per tier (at the tier's largest
// modulus), input bound (shift 0: centred values or the SKIP mod-down's 0.5 rho p, c0 = 0.625; shift 1: canonical residues of a neighbouring
// modulus, c0 = rho = 1.25), stage count, and consumer (down = 1: the mod-down epilogue subtracts the un-reduced tail from an un-reduced
// accumulator, tail <= 1 / a - MAC_FOLD_ACC; down = 0: mac_fold or a final range reduction, which only need the tail below 2^53).
// N = 16384, top tier: 18 instead of 24 instructions per butterfly column for a mod-up transform, 21 instead of 24 for a mod-down one
// (-2.5 % of a keyswitch's instructions); period-6 tier 6 instead of 12; period-12 tier 3 instead of 6.
// Encoding: two bits per global stage s (1-based) at bits 2 (s - 1): 0 N, 1 X, 2 F; bit 31 = entry present. The table is DATA: xsched_ok
// below replays the recurrence for every entry under static_assert, and tests/cpp/f64_selftest.cpp replays whole transforms on these
// schedules against exact integers at the tier tops.
struct XSchedEntry { int period, shift, down, stages; unsigned mask; };
constexpr XSchedEntry XSCHED_TABLE[] = {
    { 3, 0, 0,  8, 0x80001110u},   // NNXNXNXN          9 (periodic: 12)  3.701 of 3.969
    { 3, 0, 0,  9, 0x80004440u},   // NNNXNXNXN         9 (periodic: 12)  3.898 of 3.969
    { 3, 0, 0, 10, 0x80011140u},   // NNNXXNXNXN       12 (periodic: 18)  3.773 of 3.969
    { 3, 0, 0, 11, 0x80044440u},   // NNNXNXNXNXN      12 (periodic: 18)  3.908 of 3.969
    { 3, 0, 0, 12, 0x80111140u},   // NNNXXNXNXNXN     15 (periodic: 18)  3.809 of 3.969
    { 3, 0, 0, 13, 0x80444440u},   // NNNXNXNXNXNXN    15 (periodic: 24)  3.913 of 3.969
    { 3, 0, 0, 14, 0x81111440u},   // NNNXNXXNXNXNXN   18 (periodic: 24)  3.843 of 3.969
    { 3, 0, 0, 15, 0x84444440u},   // NNNXNXNXNXNXNXN  18 (periodic: 24)  3.916 of 3.969
    { 3, 0, 1,  8, 0x80004510u},   // NNXNXXNX         12 (periodic: 12)  3.115 of 3.969
    { 3, 0, 1,  9, 0x80011440u},   // NNNXNXXNX        12 (periodic: 12)  3.843 of 3.969
    { 3, 0, 1, 10, 0x80045110u},   // NNXNXNXXNX       15 (periodic: 18)  3.500 of 3.969
    { 3, 0, 1, 11, 0x80114440u},   // NNNXNXNXXNX      15 (periodic: 18)  3.879 of 3.969
    { 3, 0, 1, 12, 0x80451110u},   // NNXNXNXNXXNX     18 (periodic: 18)  3.701 of 3.969
    { 3, 0, 1, 13, 0x81144440u},   // NNNXNXNXNXXNX    18 (periodic: 24)  3.898 of 3.969
    { 3, 0, 1, 14, 0x84511140u},   // NNNXXNXNXNXXNX   21 (periodic: 24)  3.773 of 3.969
    { 3, 0, 1, 15, 0x91444440u},   // NNNXNXNXNXNXXNX  21 (periodic: 24)  3.908 of 3.969
    { 3, 1, 0,  8, 0x80001110u},   // NNXNXNXN          9 (periodic: 12)  3.868 of 3.969
    { 3, 1, 0,  9, 0x80004450u},   // NNXXNXNXN        12 (periodic: 18)  3.697 of 3.969
    { 3, 1, 0, 10, 0x80011110u},   // NNXNXNXNXN       12 (periodic: 18)  3.892 of 3.969
    { 3, 1, 0, 11, 0x80044510u},   // NNXNXXNXNXN      15 (periodic: 18)  3.733 of 3.969
    { 3, 1, 0, 12, 0x80111110u},   // NNXNXNXNXNXN     15 (periodic: 24)  3.905 of 3.969
    { 3, 1, 0, 13, 0x80444510u},   // NNXNXXNXNXNXN    18 (periodic: 24)  3.808 of 3.969
    { 3, 1, 0, 14, 0x81111110u},   // NNXNXNXNXNXNXN   18 (periodic: 24)  3.911 of 3.969
    { 3, 1, 0, 15, 0x84445110u},   // NNXNXNXXNXNXNXN  21 (periodic: 30)  3.822 of 3.969
    { 3, 1, 1,  8, 0x80004510u},   // NNXNXXNX         12 (periodic: 12)  3.733 of 3.969
    { 3, 1, 1,  9, 0x80011444u},   // NXNXNXXNX        15 (periodic: 18)  3.459 of 3.969
    { 3, 1, 1, 10, 0x80045110u},   // NNXNXNXXNX       15 (periodic: 18)  3.822 of 3.969
    { 3, 1, 1, 11, 0x80114450u},   // NNXXNXNXXNX      18 (periodic: 18)  3.562 of 3.969
    { 3, 1, 1, 12, 0x80451110u},   // NNXNXNXNXXNX     18 (periodic: 24)  3.868 of 3.969
    { 3, 1, 1, 13, 0x81144450u},   // NNXXNXNXNXXNX    21 (periodic: 24)  3.697 of 3.969
    { 3, 1, 1, 14, 0x84511110u},   // NNXNXNXNXNXXNX   21 (periodic: 24)  3.892 of 3.969
    { 3, 1, 1, 15, 0x91444510u},   // NNXNXXNXNXNXXNX  24 (periodic: 30)  3.733 of 3.969
    { 6, 0, 0,  8, 0x80000100u},   // NNNNXNNN          3 (periodic:  6)  4.691 of 8.000
    { 6, 0, 0,  9, 0x80000400u},   // NNNNNXNNN         3 (periodic:  6)  5.106 of 8.000
    { 6, 0, 0, 10, 0x80000400u},   // NNNNNXNNNN        3 (periodic:  6)  6.529 of 8.000
    { 6, 0, 0, 11, 0x80001000u},   // NNNNNNXNNNN       3 (periodic:  6)  7.072 of 8.000
    { 6, 0, 0, 12, 0x80010100u},   // NNNNXNNNXNNN      6 (periodic:  6)  4.946 of 8.000
    { 6, 0, 0, 13, 0x80040400u},   // NNNNNXNNNXNNN     6 (periodic: 12)  5.106 of 8.000
    { 6, 0, 0, 14, 0x80040400u},   // NNNNNXNNNXNNNN    6 (periodic: 12)  6.517 of 8.000
    { 6, 0, 0, 15, 0x80101000u},   // NNNNNNXNNNXNNNN   6 (periodic: 12)  6.688 of 8.000
    { 6, 0, 1,  8, 0x80000100u},   // NNNNXNNN          3 (periodic:  6)  4.691 of 8.000
    { 6, 0, 1,  9, 0x80000400u},   // NNNNNXNNN         3 (periodic:  6)  5.106 of 8.000
    { 6, 0, 1, 10, 0x80001000u},   // NNNNNNXNNN        3 (periodic:  6)  6.564 of 8.000
    { 6, 0, 1, 11, 0x80004100u},   // NNNNXNNXNNN       6 (periodic:  6)  4.582 of 8.000
    { 6, 0, 1, 12, 0x80010100u},   // NNNNXNNNXNNN      6 (periodic:  6)  4.946 of 8.000
    { 6, 0, 1, 13, 0x80040400u},   // NNNNNXNNNXNNN     6 (periodic: 12)  5.106 of 8.000
    { 6, 0, 1, 14, 0x80100400u},   // NNNNNXNNNNXNNN    6 (periodic: 12)  6.529 of 8.000
    { 6, 0, 1, 15, 0x80401000u},   // NNNNNNXNNNNXNNN   6 (periodic: 12)  7.072 of 8.000
    { 6, 1, 0,  8, 0x80000100u},   // NNNNXNNN          3 (periodic:  6)  5.122 of 8.000
    { 6, 1, 0,  9, 0x80000100u},   // NNNNXNNNN         3 (periodic:  6)  6.534 of 8.000
    { 6, 1, 0, 10, 0x80000400u},   // NNNNNXNNNN        3 (periodic:  6)  7.079 of 8.000
    { 6, 1, 0, 11, 0x80004040u},   // NNNXNNNXNNN       6 (periodic:  6)  4.948 of 8.000
    { 6, 1, 0, 12, 0x80010100u},   // NNNNXNNNXNNN      6 (periodic: 12)  5.122 of 8.000
    { 6, 1, 0, 13, 0x80010100u},   // NNNNXNNNXNNNN     6 (periodic: 12)  6.519 of 8.000
    { 6, 1, 0, 14, 0x80040400u},   // NNNNNXNNNXNNNN    6 (periodic: 12)  6.690 of 8.000
    { 6, 1, 0, 15, 0x80100400u},   // NNNNNXNNNNXNNNN   6 (periodic: 12)  7.264 of 8.000
    { 6, 1, 1,  8, 0x80000100u},   // NNNNXNNN          3 (periodic:  6)  5.122 of 8.000
    { 6, 1, 1,  9, 0x80000400u},   // NNNNNXNNN         3 (periodic:  6)  6.582 of 8.000
    { 6, 1, 1, 10, 0x80001040u},   // NNNXNNXNNN        6 (periodic:  6)  4.583 of 8.000
    { 6, 1, 1, 11, 0x80004040u},   // NNNXNNNXNNN       6 (periodic:  6)  4.948 of 8.000
    { 6, 1, 1, 12, 0x80010100u},   // NNNNXNNNXNNN      6 (periodic: 12)  5.122 of 8.000
    { 6, 1, 1, 13, 0x80040100u},   // NNNNXNNNNXNNN     6 (periodic: 12)  6.534 of 8.000
    { 6, 1, 1, 14, 0x80100400u},   // NNNNNXNNNNXNNN    6 (periodic: 12)  7.079 of 8.000
    { 6, 1, 1, 15, 0x80404040u},   // NNNXNNNXNNNXNNN   9 (periodic: 12)  5.027 of 8.000
    {12, 0, 0,  8, 0x80000000u},   // NNNNNNNN          0 (periodic:  0)  6.870 of 16.000
    {12, 0, 0,  9, 0x80000000u},   // NNNNNNNNN         0 (periodic:  0)  8.014 of 16.000
    {12, 0, 0, 10, 0x80000000u},   // NNNNNNNNNN        0 (periodic:  0)  9.265 of 16.000
    {12, 0, 0, 11, 0x80000000u},   // NNNNNNNNNNN       0 (periodic:  0)  10.634 of 16.000
    {12, 0, 0, 12, 0x80000000u},   // NNNNNNNNNNNN      0 (periodic:  0)  12.131 of 16.000
    {12, 0, 0, 13, 0x80000000u},   // NNNNNNNNNNNNN     0 (periodic:  6)  13.768 of 16.000
    {12, 0, 0, 14, 0x80000000u},   // NNNNNNNNNNNNNN    0 (periodic:  6)  15.559 of 16.000
    {12, 0, 0, 15, 0x80010000u},   // NNNNNNNNXNNNNNN   3 (periodic:  6)  6.870 of 16.000
    {12, 0, 1,  8, 0x80000000u},   // NNNNNNNN          0 (periodic:  0)  6.870 of 16.000
    {12, 0, 1,  9, 0x80000000u},   // NNNNNNNNN         0 (periodic:  0)  8.014 of 16.000
    {12, 0, 1, 10, 0x80000000u},   // NNNNNNNNNN        0 (periodic:  0)  9.265 of 16.000
    {12, 0, 1, 11, 0x80000000u},   // NNNNNNNNNNN       0 (periodic:  0)  10.634 of 16.000
    {12, 0, 1, 12, 0x80000000u},   // NNNNNNNNNNNN      0 (periodic:  0)  12.131 of 16.000
    {12, 0, 1, 13, 0x80000000u},   // NNNNNNNNNNNNN     0 (periodic:  6)  13.768 of 16.000
    {12, 0, 1, 14, 0x80004000u},   // NNNNNNNXNNNNNN    3 (periodic:  6)  6.444 of 16.000
    {12, 0, 1, 15, 0x80010000u},   // NNNNNNNNXNNNNNN   3 (periodic:  6)  6.870 of 16.000
    {12, 1, 0,  8, 0x80000000u},   // NNNNNNNN          0 (periodic:  0)  8.150 of 16.000
    {12, 1, 0,  9, 0x80000000u},   // NNNNNNNNN         0 (periodic:  0)  9.414 of 16.000
    {12, 1, 0, 10, 0x80000000u},   // NNNNNNNNNN        0 (periodic:  0)  10.796 of 16.000
    {12, 1, 0, 11, 0x80000000u},   // NNNNNNNNNNN       0 (periodic:  0)  12.309 of 16.000
    {12, 1, 0, 12, 0x80000000u},   // NNNNNNNNNNNN      0 (periodic:  6)  13.962 of 16.000
    {12, 1, 0, 13, 0x80000000u},   // NNNNNNNNNNNNN     0 (periodic:  6)  15.771 of 16.000
    {12, 1, 0, 14, 0x80004000u},   // NNNNNNNXNNNNNN    3 (periodic:  6)  6.994 of 16.000
    {12, 1, 0, 15, 0x80004000u},   // NNNNNNNXNNNNNNN   3 (periodic:  6)  7.754 of 16.000
    {12, 1, 1,  8, 0x80000000u},   // NNNNNNNN          0 (periodic:  0)  8.150 of 16.000
    {12, 1, 1,  9, 0x80000000u},   // NNNNNNNNN         0 (periodic:  0)  9.414 of 16.000
    {12, 1, 1, 10, 0x80000000u},   // NNNNNNNNNN        0 (periodic:  0)  10.796 of 16.000
    {12, 1, 1, 11, 0x80000000u},   // NNNNNNNNNNN       0 (periodic:  0)  12.309 of 16.000
    {12, 1, 1, 12, 0x80000000u},   // NNNNNNNNNNNN      0 (periodic:  6)  13.962 of 16.000
    {12, 1, 1, 13, 0x80001000u},   // NNNNNNXNNNNNN     3 (periodic:  6)  6.462 of 16.000
    {12, 1, 1, 14, 0x80004000u},   // NNNNNNNXNNNNNN    3 (periodic:  6)  6.994 of 16.000
    {12, 1, 1, 15, 0x80004000u},   // NNNNNNNXNNNNNNN   3 (periodic:  6)  7.754 of 16.000
};
constexpr int XSCHED_ENTRIES = sizeof(XSCHED_TABLE) / sizeof(XSCHED_TABLE[0]);
constexpr unsigned XSCHED_PRESENT = 0x80000000u;
// the schedule for a `stages`-stage forward transform (0 = none in the table: callers fall back to the periodic schedule)
HX_HD constexpr unsigned xsched_mask(int period, int shift, bool down, int stages) {
    for (int i = 0; i < XSCHED_ENTRIES; ++i)
        if (XSCHED_TABLE[i].period == period && XSCHED_TABLE[i].shift == shift && XSCHED_TABLE[i].down == (down ? 1 : 0) &&
            XSCHED_TABLE[i].stages == stages) return XSCHED_TABLE[i].mask;
    return 0u;
}
HX_HD constexpr int xsched_op(unsigned mask, int s) { return int((mask >> (2 * (s - 1))) & 3u); }     // s = global stage, 1-based (s <= 15)
namespace bounds {
constexpr double xsched_c0(int shift) { return shift ? LAZY_SKIP_MAX_RATIO : 0.5 * LAZY_SKIP_MAX_RATIO; }
constexpr bool xsched_ok(const XSchedEntry& e) {
    if (!(e.mask & XSCHED_PRESENT) || e.stages > 15) return false;                // (15 stages x 2 bits + the present bit)
    const double a = tier_a(e.period), limit = 1.0 / a, r = 0.5 + SLOP;
    double c = xsched_c0(e.shift);
    for (int s = 1; s <= e.stages; ++s) {
        const int op = xsched_op(e.mask, s);
        if (op == 0) c = (1.0 + 1.5 * a) * c + 0.5;
        else if (op == 1) c = r + 0.5 + 1.5 * a * c;
        else if (op == 2) c = r + 0.5 + 1.5 * a * r;
        else return false;
        if (!(c < limit)) return false;
    }
    if (e.down && !(MAC_FOLD_ACC + c < limit)) return false;
    // mac_fold on the un-reduced tail: |x k| < 2^103 (|l| <= 2^49) with |k| <= p / 2
    if (!(c * tier_top(e.period) * tier_top(e.period) * 0.5 < 10141204801825835211973625643008.0)) return false;
    return true;
}
constexpr bool xsched_table_ok() {
    for (int i = 0; i < XSCHED_ENTRIES; ++i)
        if (!xsched_ok(XSCHED_TABLE[i])) return false;
    return true;
}
static_assert(xsched_table_ok(), "an X schedule of XSCHED_TABLE passes 2^53 (or its tail does not fit under the mod-down epilogue): regenerate with tools/gen_xsched.py");
}  // namespace bounds

// ---- "I schedules" (round 6): the inverse transforms' range reductions by butterfly history ---------------------------------------------
// Gentleman-Sande without the w/p table: s = X + Y, d = X - Y, X' = s, Y' = mul_mod(d, w). With both inputs bounded by b p the sum and the
// difference are bounded by 2 b p (both must stay below 2^53 = p / a) and |Y'| <= (0.5 + 1.5 a 2 b) p. The two inputs of a butterfly of
// stage k + 1 have the same history -- both are sum outputs ("S") or both product outputs ("P") of stage k -- and inside a register pass
// that history is a bit of the register index: a compile-time property of the butterfly. The periodic schedule above reduces EVERY sum
// (3 instructions per butterfly and stage, plus the products of INV_NOWP_STRICT_STAGE) whatever the tier; an I schedule decides per
// stage and history whether the sum and / or the product is reduced (at the first stage of a pass the history is a bit of the THREAD
// index, so there the decision is the same for both kinds). tools/gen_isched.py brute-forces the cheapest schedule per tier (at its
// largest modulus) and geometry (LOGN, LOGE: which stages open a pass), inputs taken as canonical words of a neighbouring modulus
// (|x| <= LAZY_SKIP_MAX_RATIO p, which covers centred inputs); the last stage (n^-1 folded in, mul_shoup + reduce on both outputs)
// is not part of the schedule and only needs its sum and difference below 2^53. N = 16384: 31.5 instead of 42 + 3 reduction instructions
// per butterfly column in the top tier, 15 in the period-6 tier, 9 in the period-12 tier.
// Encoding: four bits per global stage s (1-based, s < LOGN) at bits 4 (s - 1): 1 = reduce the sum of S-S butterflies, 2 = their product,
// 4 = the sum of P-P butterflies, 8 = their product (comments: s p S P); bit 63 = entry present. bounds::isched_ok replays the recurrence
// for every entry under static_assert; tests/cpp/f64_selftest.cpp replays whole transforms against exact integers.
struct ISchedEntry { int period, logn, loge; unsigned long long mask; };
constexpr ISchedEntry ISCHED_TABLE[] = {
    { 3, 10, 4, 0x80000001c5454545ull},   // sS S sS S sS S sS PS s   21 (30)  3.871 of 3.969
    { 3, 11, 4, 0x8000001c54545545ull},   // sS S sS sS S sS S sS PS s   24 (33)  3.926 of 3.969
    { 3, 11, 5, 0x8000004510f45455ull},   // sS sS S sS S sPSp . s sS S   24 (33)  3.762 of 3.969
    { 3, 12, 4, 0x8000010f45454545ull},   // sS S sS S sS S sS S sPSp . s   25.5 (36)  3.926 of 3.969
    { 3, 12, 5, 0x800004510f454545ull},   // sS S sS S sS S sPSp . s sS S   25.5 (36)  3.798 of 3.969
    { 3, 13, 4, 0x8000450d54545455ull},   // sS sS S sS S sS S sS sPS . sS S   28.5 (39)  3.926 of 3.969
    { 3, 13, 5, 0x80004510f4545545ull},   // sS S sS sS S sS S sPSp . s sS S   28.5 (39)  3.871 of 3.969
    { 3, 14, 4, 0x8005454d45454545ull},   // sS S sS S sS S sS S sPS S sS S sS   31.5 (42)  3.926 of 3.969
    { 6, 10, 4, 0x8000000010451010ull},   // . s . s sS S . s .   9 (30)  7.750 of 8.000
    { 6, 11, 4, 0x8000000104505010ull},   // . s . sS . sS S . s .   10.5 (33)  7.812 of 8.000
    { 6, 11, 5, 0x8000001010451050ull},   // . sS . s sS S . s . s   12 (33)  7.000 of 8.000
    { 6, 12, 4, 0x8000001051051010ull},   // . s . s sS . s sS . s .   12 (36)  7.823 of 8.000
    { 6, 12, 5, 0x8000010104511010ull},   // . s . s s sS S . s . s   12 (36)  7.812 of 8.000
    { 6, 13, 4, 0x8000010451010450ull},   // . sS S . s . s sS S . s .   13.5 (39)  7.117 of 8.000
    { 6, 13, 5, 0x8000101045105010ull},   // . s . sS . s sS S . s . s   13.5 (39)  7.812 of 8.000
    { 6, 14, 4, 0x8000105105111010ull},   // . s . s s s sS . s sS . s .   15 (42)  7.945 of 8.000
    {12, 10, 4, 0x8000000010100500ull},   // . . sS . . s . s .   6 (30)  11.500 of 16.000
    {12, 11, 4, 0x8000000050100100ull},   // . . s . . s . sS . .   6 (33)  15.500 of 16.000
    {12, 11, 5, 0x8000000010100100ull},   // . . s . . s . s . .   4.5 (33)  15.500 of 16.000
    {12, 12, 4, 0x8000010010100100ull},   // . . s . . s . s . . s   6 (36)  15.500 of 16.000
    {12, 12, 5, 0x8000001001010010ull},   // . s . . s . s . . s .   6 (36)  14.305 of 16.000
    {12, 13, 4, 0x8000010100500100ull},   // . . s . . sS . . s . s .   7.5 (39)  15.625 of 16.000
    {12, 13, 5, 0x8000010010100100ull},   // . . s . . s . s . . s .   6 (39)  15.500 of 16.000
    {12, 14, 4, 0x8000050010100500ull},   // . . sS . . s . s . . sS . .   9 (42)  14.676 of 16.000
};
constexpr int ISCHED_ENTRIES = sizeof(ISCHED_TABLE) / sizeof(ISCHED_TABLE[0]);
constexpr unsigned long long ISCHED_PRESENT = 0x8000000000000000ull;
HX_HD constexpr unsigned long long isched_mask(int period, int logn, int loge) {      // 0 = none: the periodic schedule
    for (int i = 0; i < ISCHED_ENTRIES; ++i)
        if (ISCHED_TABLE[i].period == period && ISCHED_TABLE[i].logn == logn && ISCHED_TABLE[i].loge == loge) return ISCHED_TABLE[i].mask;
    return 0ull;
}
HX_HD constexpr int isched_bits(unsigned long long mask, int s) { return int((mask >> (4 * (s - 1))) & 15ull); }
// does global stage s (1-based) open a register pass of the inverse transform of this geometry? (partial pass first: KL = LOGN - (P - 1) LOGE stages)
HX_HD constexpr bool isched_opens_pass(int logn, int loge, int s) {
    const int passes = (logn + loge - 1) / loge, kl = logn - (passes - 1) * loge;
    return s == 1 || (s > kl && (s - kl - 1) % loge == 0);
}
namespace bounds {
constexpr bool isched_ok(const ISchedEntry& e) {
    if (!(e.mask & ISCHED_PRESENT) || e.logn > 15) return false;
    const double a = tier_a(e.period), limit = 1.0 / a, r = 0.5 + SLOP;
    double bs = LAZY_SKIP_MAX_RATIO, bp = LAZY_SKIP_MAX_RATIO;       // bounds of the sum outputs / product outputs of the previous stage
    for (int s = 1; s < e.logn; ++s) {
        const int bits = isched_bits(e.mask, s);
        if (isched_opens_pass(e.logn, e.loge, s) && ((bits & 3) != ((bits >> 2) & 3))) return false;   // history not known at compile time there
        if (!(2.0 * bs < limit) || !(2.0 * bp < limit)) return false;                                   // |X + Y|, |X - Y|
        const double s_from_s = (bits & 1) ? r : 2.0 * bs, p_from_s = (bits & 2) ? r : 0.5 + 1.5 * a * 2.0 * bs;
        const double s_from_p = (bits & 4) ? r : 2.0 * bp, p_from_p = (bits & 8) ? r : 0.5 + 1.5 * a * 2.0 * bp;
        bs = s_from_s > s_from_p ? s_from_s : s_from_p;
        bp = p_from_s > p_from_p ? p_from_s : p_from_p;
    }
    return 2.0 * bs < limit && 2.0 * bp < limit;                      // the fused last stage's sum and difference
}
constexpr bool isched_table_ok() {
    for (int i = 0; i < ISCHED_ENTRIES; ++i)
        if (!isched_ok(ISCHED_TABLE[i])) return false;
    return true;
}
static_assert(isched_table_ok(), "an I schedule of ISCHED_TABLE passes 2^53 or decides by history where a pass opens: regenerate with tools/gen_isched.py");
}  // namespace bounds

}  // namespace hxf
