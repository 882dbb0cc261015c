// ntt_core_f64.hpp -- workgroup-level negacyclic NTT / INTT on the FP64 pipe (exact, moduli < 2^52).
// Same register-pass / LDS re-deal structure and index maps as ntt_core.hpp (Geom, A/B layouts); only the
// butterflies differ (f64_arith.hpp). Values are centred residues in doubles. Canonical transforms:
//   forward : device/keyswitch/ntt_core.hpp:68-386   (twiddle index m+i, :222)
//   inverse : device/keyswitch/intt_core.hpp:103-448 + _intt_normalize :72-93 (n^-1 folded into the last
//             stage here, as the standalone inverse kernel does)
#pragma once
#include "f64_arith.hpp"
#include "ntt_core.hpp"

namespace hx {

using hxf::Mod;

// Twiddle tables are written by the host before any kernel runs and never change: reading them through the
// constant address space tells the compiler so. Without it, a kernel that also stores to global memory inside a
// loop (k_ksf_up) loses its scalar (s_load) twiddle fetches for the wave-uniform passes -- the stores might alias.
typedef const __attribute__((address_space(4))) double* ctw_t;

// LAZY = forward reduction period (0: strict; 3, 6 or 12 by modulus size, f64_arith.hpp): butterflies skip the range reduction except after every LAZY-th
// global stage and after the last one (bounds in f64_arith.hpp). LOGN is only needed to find the last stage.
// UNI: the twiddle index is wave-uniform (scalar loads through the constant address space)
// SHIFT: phase of the reduction schedule (f64_arith.hpp lazy_fwd_reduce_after; 1 = un-centred inputs taken as they are)
// NORED: global stage whose PERIODIC reduction is dropped (0 = none): on the shifted schedule the last stage of a 2^14-point
// transform is a periodic reduction point ((14 + 1) % 3 == 0); a transform whose consumer takes an un-reduced tail (FINAL = false)
// drops it and hands over three un-reduced stages, 3.45p, instead (f64_arith.hpp)
// SEMI (strict kernels only, LAZY == 0): the semi-strict schedule of f64_arith.hpp ct_bfly_semi -- Shoup-form products (the w/p table
// IS read here) and outputs reduced only where the next stage adds them; the last stage of the call reduces everything.
// XS != 0 (round 6, lazy kernels): an X schedule of f64_arith.hpp (xsched_mask) replaces the periodic one -- per global stage nothing, the
// added operand or both operands are range-reduced IN FRONT of the butterfly; SHIFT and NORED are then unused (the mask was chosen for the
// input bound and the consumer), LOGN != 0 still asks for the full reduction after the last stage.
template <int E, int OFF, int K, int S0, int LOGN = 0, int LAZY = 0, bool UNI = false, int SHIFT = 0, int NORED = 0, bool SEMI = false, unsigned XS = 0u>
__device__ __forceinline__ void fwd_stages_f64(double (&v)[E], u32 G, const double* __restrict__ w,
                                               const double* __restrict__ wp, const Mod m) {
    static_assert(XS == 0u || (LAZY > 0 && !SEMI), "X schedules: lazy kernels");
    if constexpr (SEMI) {
        static_assert(LAZY == 0, "semi-strict schedule: strict kernels");
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const u32 base = (1u << (S0 - 1 + u)) + (G << u);
#pragma unroll
            for (int j = 0; j < (1 << u); ++j) {
                const double W = UNI ? ((ctw_t)w)[base + j] : w[base + j];
                const double Wp = UNI ? ((ctw_t)wp)[base + j] : wp[base + j];
#pragma unroll
                for (int c = 0; c < (1 << (K - 1 - u)); ++c) {
                    const int a0 = OFF + (j << (K - u)) + c;
                    // the next stage pairs (a, a + 2^(K-2-u)): bit K-2-u of c says whether these two outputs are added or multiplied there
                    const bool added_next = (u == K - 1) || (((c >> (K - 2 - u)) & 1) == 0);
                    hxf::ct_bfly_semi(v[a0], v[a0 + (1 << (K - 1 - u))], W, Wp, m, added_next);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const u32 base = (1u << (S0 - 1 + u)) + (G << u);
        const bool red = XS ? (LOGN != 0 && S0 + u == LOGN)
                            : (!LAZY || (hxf::lazy_fwd_reduce_after(S0 + u, LOGN, LAZY ? LAZY : 3, SHIFT) && (S0 + u) != NORED));
        const int xop = XS ? hxf::xsched_op(XS, S0 + u) : 0;
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const double W = UNI ? ((ctw_t)w)[base + j] : w[base + j];          // forward butterflies need no w/p table
#pragma unroll
            for (int c = 0; c < (1 << (K - 1 - u)); ++c) {
                const int a0 = OFF + (j << (K - u)) + c;
                if (xop >= 1) v[a0] = hxf::reduce(v[a0], m);
                if (xop == 2) v[a0 + (1 << (K - 1 - u))] = hxf::reduce(v[a0 + (1 << (K - 1 - u))], m);
                if (red) hxf::ct_bfly(v[a0], v[a0 + (1 << (K - 1 - u))], W, m);
                else     hxf::ct_bfly_lazy(v[a0], v[a0 + (1 << (K - 1 - u))], W, m);
            }
        }
    }
}

// the same K stages with their 2^K - 1 twiddles already in registers (tw[(1 << u) - 1 + j] = stage u, sub-block j)
template <int E, int OFF, int K, int S0, int LOGN = 0, int LAZY = 0, int SHIFT = 0, int NORED = 0, unsigned XS = 0u>
__device__ __forceinline__ void fwd_stages_f64_tw(double (&v)[E], const double (&tw)[(1 << K) - 1], const Mod m) {
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const bool red = XS ? (LOGN != 0 && S0 + u == LOGN)
                            : (!LAZY || (hxf::lazy_fwd_reduce_after(S0 + u, LOGN, LAZY ? LAZY : 3, SHIFT) && (S0 + u) != NORED));
        const int xop = XS ? hxf::xsched_op(XS, S0 + u) : 0;
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const double W = tw[(1 << u) - 1 + j];
#pragma unroll
            for (int c = 0; c < (1 << (K - 1 - u)); ++c) {
                const int a0 = OFF + (j << (K - u)) + c;
                if (xop >= 1) v[a0] = hxf::reduce(v[a0], m);
                if (xop == 2) v[a0 + (1 << (K - 1 - u))] = hxf::reduce(v[a0 + (1 << (K - 1 - u))], m);
                if (red) hxf::ct_bfly(v[a0], v[a0 + (1 << (K - 1 - u))], W, m);
                else     hxf::ct_bfly_lazy(v[a0], v[a0 + (1 << (K - 1 - u))], W, m);
            }
        }
    }
}

// K per-lane stages with the twiddles of stages 0 .. K-2 (2^(K-1) - 1 of them) requested up front, those of the last stage
// where the compiler puts them: one exposed wait instead of K - 1 at the price of 2^K - 2 registers. (Requesting stage
// u + 2 ahead of the butterflies of stage u as well -- two stages' twiddles live -- spilled 24-36 registers in the
// keyswitch kernels: 179 k against 200 k keyswitch/s.)
template <int E, int OFF, int K, int S0, int LOGN = 0, int LAZY = 0, int SHIFT = 0, unsigned XS = 0u>
__device__ __forceinline__ void fwd_stages_f64_ahead(double (&v)[E], u32 G, const double* __restrict__ w, const Mod m) {
    double tw[(1 << (K - 1)) - 1];
#pragma unroll
    for (int u = 0; u + 1 < K; ++u)
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) tw[(1 << u) - 1 + j] = w[(1u << (S0 - 1 + u)) + (G << u) + j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const bool red = XS ? (LOGN != 0 && S0 + u == LOGN) : (!LAZY || hxf::lazy_fwd_reduce_after(S0 + u, LOGN, LAZY ? LAZY : 3, SHIFT));
        const int xop = XS ? hxf::xsched_op(XS, S0 + u) : 0;
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const double W = u + 1 < K ? tw[(1 << u) - 1 + j] : w[(1u << (S0 - 1 + u)) + (G << u) + j];
#pragma unroll
            for (int c = 0; c < (1 << (K - 1 - u)); ++c) {
                const int a0 = OFF + (j << (K - u)) + c;
                if (xop >= 1) v[a0] = hxf::reduce(v[a0], m);
                if (xop == 2) v[a0 + (1 << (K - 1 - u))] = hxf::reduce(v[a0 + (1 << (K - 1 - u))], m);
                if (red) hxf::ct_bfly(v[a0], v[a0 + (1 << (K - 1 - u))], W, m);
                else     hxf::ct_bfly_lazy(v[a0], v[a0 + (1 << (K - 1 - u))], W, m);
            }
        }
    }
}

using hxf::InvScale;

// one inverse butterfly of global stage `gs` (1-based); NOWP: no w/p table (f64_arith.hpp gs_bfly_lazy_nowp)
// IS != 0 (round 6, lazy kernels without the w/p table): an I schedule of f64_arith.hpp (isched_mask) decides per stage and per history of
// the two inputs -- `from_products`: both are product outputs of the previous stage (a bit of the register index inside a pass; at the
// first stage of a pass the schedule holds the same decision for both kinds, so callers pass false) -- which outputs are range-reduced
template <int LAZY, bool NOWP, unsigned long long IS = 0ull>
__device__ __forceinline__ void inv_bfly(double& X, double& Y, double W, double Wp, const Mod m, int gs, bool from_products = false) {
    if constexpr (IS != 0ull) {
        static_assert(LAZY > 0 && NOWP, "I schedules: lazy kernels without the w/p table");
        const int bits = hxf::isched_bits(IS, gs) >> (from_products ? 2 : 0);
        const double s = X + Y, d = X - Y;
        X = (bits & 1) ? hxf::reduce(s, m) : s;
        const double t = hxf::mul_mod(d, W, m);
        Y = (bits & 2) ? hxf::reduce(t, m) : t;
    } else if constexpr (NOWP) {
        if (LAZY && gs != hxf::INV_NOWP_STRICT_STAGE) hxf::gs_bfly_lazy_nowp(X, Y, W, m);
        else hxf::gs_bfly_nowp(X, Y, W, m);
    } else {
        if (LAZY) hxf::gs_bfly_lazy(X, Y, W, Wp, m);
        else      hxf::gs_bfly(X, Y, W, Wp, m);
    }
}

template <int E, int OFF, int K, int LO, int LOGN, bool LAST, int LAZY = 0, bool UNI = false, bool NOWP = false, unsigned long long IS = 0ull>
__device__ __forceinline__ void inv_stages_f64(double (&v)[E], u32 G, const double* __restrict__ iw,
                                               const double* __restrict__ iwp, const Mod m, const InvScale sc) {
    constexpr u32 N = 1u << LOGN;
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const bool fused = LAST && (u == K - 1);
        const u32 base = N - (N >> (LO + u)) + 1 + (G << (K - 1 - u));
#pragma unroll
        for (int j = 0; j < (1 << (K - 1 - u)); ++j) {
            double W = 0, Wp = 0;
            if (!fused) {
                W = UNI ? ((ctw_t)iw)[base + j] : iw[base + j];
                if (!NOWP) Wp = UNI ? ((ctw_t)iwp)[base + j] : iwp[base + j];
            }
#pragma unroll
            for (int c = 0; c < (1 << u); ++c) {
                const int a0 = OFF + (j << (u + 1)) + c;
                const int a1 = a0 + (1 << u);
                if (!fused) {
                    inv_bfly<LAZY, NOWP, IS>(v[a0], v[a1], W, Wp, m, LO + u + 1, u > 0 && ((c >> (u > 0 ? u - 1 : 0)) & 1));
                } else {                                   // last stage: scale both outputs by n^-1
                    const double s = v[a0] + v[a1], d = v[a0] - v[a1];
                    v[a0] = hxf::reduce(hxf::mul_shoup(s, sc.n, sc.n_p, m), m);
                    v[a1] = hxf::reduce(hxf::mul_shoup(d, sc.nw, sc.nw_p, m), m);
                }
            }
        }
    }
}

// The same K inverse stages with all their 2^K - 1 twiddle pairs requested before the first butterfly (IPRE, kernels with
// registers to spare: k_ksx_intt). Left alone the compiler requests each pair right in front of its use and waits for it:
// up to 15 exposed latencies per per-lane pass.
template <int E, int OFF, int K, int LO, int LOGN, bool LAST, int LAZY = 0, bool NOWP = false, unsigned long long IS = 0ull>
__device__ __forceinline__ void inv_stages_f64_pre(double (&v)[E], u32 G, const double* __restrict__ iw,
                                                   const double* __restrict__ iwp, const Mod m, const InvScale sc) {
    constexpr u32 N = 1u << LOGN;
    constexpr int NT = (1 << K) - 1;
    double tw[NT], twp[NOWP ? 1 : NT];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const bool fused = LAST && (u == K - 1);
        const u32 base = N - (N >> (LO + u)) + 1 + (G << (K - 1 - u));
#pragma unroll
        for (int j = 0; j < (1 << (K - 1 - u)); ++j)
            if (!fused) {
                tw[(1 << K) - (1 << (K - u)) + j] = iw[base + j];
                if constexpr (!NOWP) twp[(1 << K) - (1 << (K - u)) + j] = iwp[base + j];
            }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const bool fused = LAST && (u == K - 1);
#pragma unroll
        for (int j = 0; j < (1 << (K - 1 - u)); ++j) {
            const double W = fused ? 0.0 : tw[(1 << K) - (1 << (K - u)) + j];
            double Wp = 0.0;
            if constexpr (!NOWP) Wp = fused ? 0.0 : twp[(1 << K) - (1 << (K - u)) + j];
#pragma unroll
            for (int c = 0; c < (1 << u); ++c) {
                const int a0 = OFF + (j << (u + 1)) + c;
                const int a1 = a0 + (1 << u);
                if (!fused) {
                    inv_bfly<LAZY, NOWP, IS>(v[a0], v[a1], W, Wp, m, LO + u + 1, u > 0 && ((c >> (u > 0 ? u - 1 : 0)) & 1));
                } else {
                    const double s = v[a0] + v[a1], d = v[a0] - v[a1];
                    v[a0] = hxf::reduce(hxf::mul_shoup(s, sc.n, sc.n_p, m), m);
                    v[a1] = hxf::reduce(hxf::mul_shoup(d, sc.nw, sc.nw_p, m), m);
                }
            }
        }
    }
}

template <class G, class FromIdx, class ToIdx>
__device__ __forceinline__ void redeal_f64(double (&v)[G::E], double* lds, int tid, FromIdx from, ToIdx to) {
#pragma unroll
    for (int r = 0; r < G::E; ++r) lds[G::pad(from(r, tid))] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = lds[G::pad(to(r, tid))];
    __syncthreads();
}

// PRE > 0 (register-tight kernels, keyswitch_x.hip): the per-lane twiddles of the first PRE % 10 groups of the LAST (partial)
// pass are requested before the re-deal that precedes it, those of the other groups right behind it, ahead of all butterflies
// (194 k -> 200 k keyswitch/s); PRE >= 10: the per-lane full pass requests its early stages' twiddles up front as well
// (fwd_stages_f64_ahead: -> 202 k). Left alone, a kernel that holds 96 data registers gets every twiddle load of that pass right in
// front of its first use with an s_waitcnt vmcnt(0) behind it -- eight fully exposed L2 latencies per transform.
// Wave priority by pass of a forward transform: four decimal digits d0 d1 d2 d3, pass k runs at s_setprio(dk - 1) (0 = off);
// whatever follows the transform inherits the last pass's priority. HX_FWD_PRIO is the translation unit's default for WgNttF64's
// FPRIO parameter. 1222 -- the pass in front of the cross-wave barrier below everything else -- pays in kernels whose workgroups
// run transform after transform (keyswitch_x.hip has the reasoning and the numbers); a workgroup that runs ONE transform loses a
// few per cent with it (k_ksf_ntt_up / k_ksf_moddown at batch 32: -4 %), so it is chosen per kernel.
#ifndef HX_FWD_PRIO
#define HX_FWD_PRIO 0
#endif
template <int PASS, int KNOB = HX_FWD_PRIO>
__device__ __forceinline__ void hx_fwd_prio() {
    if constexpr (KNOB != 0 && PASS < 4) {
        constexpr int d = PASS == 0 ? KNOB / 1000 : PASS == 1 ? (KNOB / 100) % 10 : PASS == 2 ? (KNOB / 10) % 10 : KNOB % 10;
        __builtin_amdgcn_s_setprio(d - 1);
    }
}
// ... and of an inverse transform: HX_INV_PRIO digits = partial first pass, the two wave-private passes, the pass behind the
// cross-wave re-deal (the geometry of N = 16384: four passes)
#ifndef HX_INV_PRIO
#define HX_INV_PRIO 0
#endif
template <int PH>
__device__ __forceinline__ void hx_inv_prio() {
    if constexpr (HX_INV_PRIO != 0 && PH < 4) {
        constexpr int d = PH == 0 ? HX_INV_PRIO / 1000 : PH == 1 ? (HX_INV_PRIO / 100) % 10 : PH == 2 ? (HX_INV_PRIO / 10) % 10 : HX_INV_PRIO % 10;
        __builtin_amdgcn_s_setprio(d - 1);
    }
}
// FSHIFT: phase of the forward reduction schedule (1: un-centred inputs, f64_arith.hpp); NOWP: inverse transforms without
// the w/p table
// TOP > 0 (round 5: the slot-major keyswitch at N = 32768): this workgroup transforms ONE of the 2^TOP blocks of a 2^(LOGN + TOP)-point
// transform -- the inner LOGN stages, i.e. global forward stages TOP + 1 ... TOP + LOGN (global inverse stages 1 ... LOGN), with the stage
// numbers (reduction schedule), table size and twiddle group indices of the FULL transform; `top` (wave-uniform) = which block. The
// outer TOP stages are the caller's: a radix-2 step across the blocks on load / in a finishing pass (keyswitch_x.hip HALF kernels), the
// same cut keyswitch_lat.hip makes in four (WgSubNtt).
// SEMIU (strict kernels, round 5): the semi-strict schedule in the passes whose twiddles are WAVE-UNIFORM only (the first pass and every
// pass with LO >= 6: for N = 16384 eight of the fourteen stages) -- there w and w/p come through the scalar cache, so the second table costs
// no vector loads, no registers and does not disturb the per-lane passes' early twiddle requests (PRE), which is what made the all-passes
// variant of round 4 lose. The schedule is pass-local (f64_arith.hpp ct_bfly_semi: a pass starts from reduced values and its last stage
// reduces everything), so strict and semi-strict passes mix freely.
// XSD (round 6, lazy kernels): -1 = the periodic reduction schedule; 0 / 1 = the X schedule of f64_arith.hpp for this tier, input kind
// (FSHIFT) and transform size, chosen for a consumer that takes any tail below 2^53 (mac_fold, or FINAL's range reduction: 0) or for the
// mod-down epilogue (un-reduced accumulator minus the tail: 1). Strict kernels keep their schedules.
// ISD (round 6, lazy kernels without the w/p table, TOP == 0): the inverse transform on the I schedule of f64_arith.hpp for this tier and geometry
// (none in the table: the periodic schedule)
template <int LOGN, int LOGE, int LAZY = 0, int PRE = 0, int FSHIFT = 0, bool NOWP = false, int FPRIO = HX_FWD_PRIO, int TOP = 0, bool SEMIU = false, int XSD = -1,
          bool ISD = false>
struct WgNttF64 {
    static constexpr unsigned long long IS = (ISD && LAZY > 0 && NOWP && TOP == 0) ? hxf::isched_mask(LAZY, LOGN, LOGE) : 0ull;
    static_assert(!SEMIU || LAZY == 0, "semi-strict uniform passes: strict kernels");
    using G = Geom<LOGN, LOGE>;
    static constexpr int E = G::E;
    static constexpr int FLOGN = LOGN + TOP;                      // log2 of the full transform
    static constexpr unsigned XS = (XSD >= 0 && LAZY > 0) ? hxf::xsched_mask(LAZY, FSHIFT, XSD == 1, FLOGN) : 0u;
    static_assert(XSD < 0 || LAZY <= 0 || XS != 0u, "no X schedule in XSCHED_TABLE for this tier / input / size");
    // (TOP > 0: the caller's outer stages follow the same mask -- keyswitch_x.hip ksh_combine, ntt.hip k_ntt_fwd_h: stage 1 is an N stage in every 15-stage schedule)
    static_assert(TOP == 0 || XS == 0u || (TOP == 1 && hxf::xsched_op(XS, 1) == 0), "the outer stage of a split transform must be an N stage of its X schedule");
    // group index of the full transform for a local one at the pass whose first LOCAL stage is S0L (forward) ...
    template <int S0L>
    __device__ static __forceinline__ u32 gfwd(u32 top, u32 Gl) { if constexpr (TOP > 0) return (top << (S0L - 1)) | Gl; else return Gl; }
    // ... and for an inverse call that covers local stages LO + 1 ... LO + K
    template <int LO, int K>
    __device__ static __forceinline__ u32 ginv(u32 top, u32 Gl) { if constexpr (TOP > 0) return (top << (LOGN - LO - K)) | Gl; else return Gl; }

    // forward: A layout in, B layout out, all values centred. FRESH as in ntt_core.hpp (single-transform kernels).
    // FINAL = false (LAZY only): the range reduction after the last stage is left to the consumer, outputs are
    // then bounded by 2.14p instead of p/2 (two unreduced stages after the reduction at the last multiple of three;
    // f64_arith.hpp) -- mul_mod and mul_shoup of the mod-up / mod-down epilogues accept that.
    // `after_cross` runs right after the first (cross-wave) re-deal: the place to request data the epilogue will
    // need (the barriers and fences of the re-deals pin every load the compiler sees behind them).
    struct NoHook { __device__ __forceinline__ void operator()() const {} };
    // `before_last` runs between the last re-deal and the partial pass (keyswitch_x.hip requests the first keys of the
    // multiply-accumulate there).
    template <int PASS, bool FRESH = false, bool FINAL = true, class Hook = NoHook, class Hook2 = NoHook>
    __device__ static __forceinline__ void fwd_pass(double (&v)[E], double* lds, int tid, const double* w,
                                                    const double* wp, const Mod m, Hook after_cross = Hook(),
                                                    Hook2 before_last = Hook2(), u32 top = 0) {
        hx_fwd_prio<PASS, FPRIO>();
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = LOGN - (PASS + 1) * LOGE;
            // LO >= 6: every lane of a wave shares the group index -> scalar twiddle loads
            const u32 Gl = (PASS == 0) ? 0u : (LO >= 6 ? u32(__builtin_amdgcn_readfirstlane(u32(tid) >> LO)) : (u32(tid) >> LO));
            const u32 Gp = gfwd<PASS * LOGE + 1>(top, Gl);
            if constexpr (PRE >= 10 && !(PASS == 0 || LO >= 6)) fwd_stages_f64_ahead<E, 0, LOGE, PASS * LOGE + 1 + TOP, FLOGN, LAZY, FSHIFT, XS>(v, Gp, w, m);
            else fwd_stages_f64<E, 0, LOGE, PASS * LOGE + 1 + TOP, FLOGN, LAZY, (PASS == 0 || LO >= 6), FSHIFT, 0,
                                (SEMIU && (PASS == 0 || LO >= 6)), XS>(v, Gp, w, wp, m);
            constexpr bool LEAD = !(FRESH && PASS == 0);
            if constexpr (PRE > 0 && PASS + 1 == G::P - 1 && PASS > 0) {
                constexpr int NT = (1 << G::KL) - 1, S0L = (G::P - 1) * LOGE + 1;
                constexpr int NPRE = (PRE % 10) < G::NG ? (PRE % 10) : G::NG;
                double tl[G::NG][NT];
                auto request = [&](int g) {
                    const u32 Gb = gfwd<S0L>(top, u32(G::grpB(g, tid)));
#pragma unroll
                    for (int u = 0; u < G::KL; ++u)
#pragma unroll
                        for (int j = 0; j < (1 << u); ++j) tl[g][(1 << u) - 1 + j] = w[(1u << (S0L + TOP - 1 + u)) + (Gb << u) + j];
                };
#pragma unroll
                for (int g = 0; g < NPRE; ++g) request(g);
                redeal_pass<G, LO, LOGE, true, LEAD, true>(v, lds, tid);
#pragma unroll
                for (int g = NPRE; g < G::NG; ++g) request(g);
                __builtin_amdgcn_sched_barrier(0);
                before_last();
                fwd_last_tw<0, FINAL>(v, tl, m);
                return;
            }
            redeal_pass<G, LO, LOGE, true, LEAD, (PASS + 1 == G::P - 1)>(v, lds, tid);
            if constexpr (PASS == 0) after_cross();
            fwd_pass<PASS + 1, FRESH, FINAL>(v, lds, tid, w, wp, m, NoHook(), before_last, top);
        } else {
            before_last();
            fwd_last<0, FINAL>(v, tid, w, wp, m, top);
        }
    }
    template <int GRP, bool FINAL = true>
    __device__ static __forceinline__ void fwd_last(double (&v)[E], int tid, const double* w, const double* wp,
                                                    const Mod m, u32 top = 0) {
        if constexpr (GRP < G::NG) {
            const u32 Gbits = gfwd<(G::P - 1) * LOGE + 1>(top, u32(G::grpB(GRP, tid)));
            // LOGN = 0 tells the stage loop that no stage is the last one
            fwd_stages_f64<E, GRP * (1 << G::KL), G::KL, (G::P - 1) * LOGE + 1 + TOP, FINAL ? FLOGN : 0, LAZY, false, FSHIFT,
                           (!FINAL && FSHIFT != 0) ? FLOGN : 0, false, XS>(v, Gbits, w, wp, m);
            fwd_last<GRP + 1, FINAL>(v, tid, w, wp, m, top);
        }
    }
    template <int GRP, bool FINAL = true>
    __device__ static __forceinline__ void fwd_last_tw(double (&v)[E], const double (&tl)[G::NG][(1 << G::KL) - 1], const Mod m) {
        if constexpr (GRP < G::NG) {
            fwd_stages_f64_tw<E, GRP * (1 << G::KL), G::KL, (G::P - 1) * LOGE + 1 + TOP, FINAL ? FLOGN : 0, LAZY, FSHIFT,
                              (!FINAL && FSHIFT != 0) ? FLOGN : 0, XS>(v, tl[GRP], m);
            fwd_last_tw<GRP + 1, FINAL>(v, tl, m);
        }
    }
    template <bool FRESH = false, bool FINAL = true, class Hook = NoHook, class Hook2 = NoHook>
    __device__ static __forceinline__ void forward(double (&v)[E], double* lds, int tid, const double* w,
                                                   const double* wp, const Mod m, Hook after_cross = Hook(),
                                                   Hook2 before_last = Hook2(), u32 top = 0) {
        static_assert(G::P > 1, "single-pass geometries are not used");
        fwd_pass<0, FRESH, FINAL>(v, lds, tid, w, wp, m, after_cross, before_last, top);
    }
    // every pass except the last (partial) one, ending with the re-deal into B layout; fwd_last<0> finishes.
    // Lets a persistent kernel slot the next polynomial's loads between the two.
    template <int PASS>
    __device__ static __forceinline__ void fwd_pass_until_last(double (&v)[E], double* lds, int tid, const double* w,
                                                               const double* wp, const Mod m) {
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = LOGN - (PASS + 1) * LOGE;
            // LO >= 6: every lane of a wave shares the group index -> scalar twiddle loads
            const u32 Gp = (PASS == 0) ? 0u : (LO >= 6 ? u32(__builtin_amdgcn_readfirstlane(u32(tid) >> LO)) : (u32(tid) >> LO));
            fwd_stages_f64<E, 0, LOGE, PASS * LOGE + 1, LOGN, LAZY, false, FSHIFT, 0, false, XS>(v, Gp, w, wp, m);
            if constexpr (PASS + 1 < G::P - 1) {
                constexpr int LO2 = LO - LOGE;
                redeal_f64<G>(v, lds, tid, [](int r, int t) { return G::template idxF<LO>(r, t); },
                              [](int r, int t) { return G::template idxF<LO2>(r, t); });
            } else {
                redeal_f64<G>(v, lds, tid, [](int r, int t) { return G::template idxF<LO>(r, t); },
                              [](int r, int t) { return G::idxB(r, t); });
            }
            fwd_pass_until_last<PASS + 1>(v, lds, tid, w, wp, m);
        }
    }

    // inverse: B layout in, A layout out, centred, scaled by n^-1
    template <int GRP, bool IPRE = false>
    __device__ static __forceinline__ void inv_first(double (&v)[E], int tid, const double* iw, const double* iwp,
                                                     const Mod m, const InvScale sc, u32 top = 0) {
        if constexpr (GRP < G::NG) {
            const u32 Gbits = ginv<0, G::KL>(top, u32(G::grpB(GRP, tid)));
            if constexpr (IPRE) inv_stages_f64_pre<E, GRP * (1 << G::KL), G::KL, 0, FLOGN, (G::P == 1 && TOP == 0), LAZY, NOWP, IS>(v, Gbits, iw, iwp, m, sc);
            else inv_stages_f64<E, GRP * (1 << G::KL), G::KL, 0, FLOGN, (G::P == 1 && TOP == 0), LAZY, false, NOWP, IS>(v, Gbits, iw, iwp, m, sc);
            inv_first<GRP + 1, IPRE>(v, tid, iw, iwp, m, sc, top);
        }
    }
    // `before_uniform` runs once, right after the re-deal that precedes the first pass whose twiddles are wave-uniform
    // (scalar loads): from there on the transform waits for no vector load, so a long-latency request issued there
    // (a persistent kernel's next input) delays nothing -- vector memory returns in order.
    template <int PASS>
    static constexpr bool inv_pass_uniform = (PASS == G::P - 2) || (G::KL + PASS * LOGE >= 6);
    template <int PASS, bool FRESH = false, class Hook = NoHook, bool IPRE = false, class Gate = NoGate>
    __device__ static __forceinline__ void inv_pass(double (&v)[E], double* lds, int tid, const double* iw,
                                                    const double* iwp, const Mod m, const InvScale sc,
                                                    Hook before_uniform = Hook(), u32 top = 0, Gate* gate = nullptr) {
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = G::KL + PASS * LOGE;
            constexpr bool LEAD = !(FRESH && PASS == 0);
            hx_inv_prio<PASS + 1>();
            // ReadersGate (ntt_core.hpp): wait in front of the first (wave-private) re-deal, arrive behind the cross-wave exchange
            if constexpr (gate_on<Gate> && PASS == 0 && !G::HALF_ONLY) gate->wait();
            redeal_pass<G, LO, LOGE, false, LEAD, (PASS == 0)>(v, lds, tid);
            if constexpr (gate_on<Gate> && PASS == G::P - 2 && !G::HALF_ONLY && !G::template wave_private<LO>) gate->arrive();
            if constexpr (inv_pass_uniform<PASS> && (PASS == 0 || !inv_pass_uniform<(PASS > 0 ? PASS - 1 : 0)>)) before_uniform();
            const u32 Gl = (PASS == G::P - 2) ? 0u : (LO >= 6 ? u32(__builtin_amdgcn_readfirstlane(u32(tid) >> LO)) : (u32(tid) >> LO));
            const u32 Gp = ginv<LO, LOGE>(top, Gl);
            // (TOP > 0: the transform's last stage -- the one with n^-1 folded in -- is not among these)
            if constexpr (IPRE && !(PASS == G::P - 2 || LO >= 6)) inv_stages_f64_pre<E, 0, LOGE, LO, FLOGN, false, LAZY, NOWP, IS>(v, Gp, iw, iwp, m, sc);
            else inv_stages_f64<E, 0, LOGE, LO, FLOGN, (PASS == G::P - 2 && TOP == 0), LAZY, (PASS == G::P - 2 || LO >= 6), NOWP, IS>(v, Gp, iw, iwp, m, sc);
            inv_pass<PASS + 1, FRESH, Hook, IPRE, Gate>(v, lds, tid, iw, iwp, m, sc, before_uniform, top, gate);
        }
    }
    template <bool FRESH = false, class Hook = NoHook, bool IPRE = false, class Gate = NoGate>
    __device__ static __forceinline__ void inverse(double (&v)[E], double* lds, int tid, const double* iw,
                                                   const double* iwp, const Mod m, const InvScale sc,
                                                   Hook before_uniform = Hook(), u32 top = 0, Gate* gate = nullptr) {
        static_assert(inverse_is_ordered<FRESH, Gate>, "inverse<false>: pass a ReadersGate, or OrderedByCaller if a forward transform or a barrier precedes");
        hx_inv_prio<0>();
        inv_first<0, IPRE>(v, tid, iw, iwp, m, sc, top);
        inv_pass<0, FRESH, Hook, IPRE, Gate>(v, lds, tid, iw, iwp, m, sc, before_uniform, top, gate);
    }
};

// Per-limb arithmetic tier (round 5). A kernel built with LAZY >= 0 runs every transform on that one schedule (plans whose moduli share a
// tier). Built with LAZY = -1 it looks the schedule up per limb: a workgroup of these kernels transforms modulo ONE q_i, so the choice is
// one wave-uniform branch around the transform (the variants it does not take cost instruction-cache space only; these are the
// small-batch kernels, one or two transforms per workgroup). WIDE: the ring dimension has kernels for all four periods (N = 16384);
// elsewhere 6 and 12 run as 3, which is always valid.
template <int V> struct TierC { static constexpr int value = V; };
template <int LAZY, bool WIDE, class F>
__device__ __forceinline__ void with_tier(unsigned long long tiermap, u32 limb, F f) {
    if constexpr (LAZY >= 0) {
        f(TierC<LAZY>{});
    } else {
        const u32 t = __builtin_amdgcn_readfirstlane(u32(tiermap >> (4 * limb)) & 15u);
        if constexpr (WIDE) {
            if (t == 12) f(TierC<12>{});
            else if (t == 6) f(TierC<6>{});
            else if (t == 3) f(TierC<3>{});
            else f(TierC<0>{});
        } else {
            if (t != 0) f(TierC<3>{});
            else f(TierC<0>{});
        }
    }
}

}  // namespace hx
