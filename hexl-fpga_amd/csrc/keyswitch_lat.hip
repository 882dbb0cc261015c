// keyswitch_lat.hip -- K4 on the FP64 pipe for a LONE keyswitch at N = 16384 (round 4): the latency path.
//
// The SEAL bridge calls KeySwitch at worksize 1 (experimental/bridge-seal/tests/fpga_context.h:13-16). Its dataflow is four
// transforms deep (INTT -> NTT -> INTT_sp -> NTT), and a 16384-point transform on ONE compute unit is 9-10 us of FP64 issue
// however it is scheduled (16 waves share 4 SIMDs): the five-kernel (b, d)-major path takes 72 us on the device, a
// three-kernel fusion of it 70 us (keyswitch_f64.hip k_ksl_*: the kernel count was never the bound). What a lone keyswitch has
// in abundance is idle compute units -- so every transform is cut in FOUR:
//
//   a 2^14-point negacyclic transform = its two outermost stages (a radix-4 step across the four quarters of the polynomial,
//   wave-uniform twiddles) + four independent 2^12-point sub-transforms on the quarters (the remaining twelve stages; quarter
//   q uses the twiddles whose group index has q on top -- WgSubNtt below).
//
// A sub-transform is 256 threads x 16 coefficients, one wave per SIMD: ~2 us instead of ~10. The outermost stages move to
// whoever consumes / produces the quarters, so no transform needs a kernel boundary of its own:
//
//   k_ksq_intt   (d, q)          twelve inverse stages on quarter q of t_target[d]                     -> sub[d] (centred doubles)
//   k_ksq_up     (slot, d, q)    the last two inverse stages (+ n^-1) on the four quarters of sub[d] = c_d, c_d mod q_slot, the first
//                                two forward stages, quarter q of that, twelve forward stages, times key[d][slot][k], canonical,
//                                ADDED into prod[k][slot] by 64-bit integer atomics (slot == d: t_d itself, no transform)
//   k_ksq_intt_sp (k, q)         twelve inverse stages on quarter q of the accumulated special limb             -> subsp[k]
//   k_ksq_down   (k, i, q)       the last two inverse stages -> y_k = s'_k - floor(q_sp/2) (exact centred remainder), the first two
//                                forward stages modulo q_i, twelve forward stages on quarter q, mod-switch epilogue on prod[k][i]
//
// 4 L + 4 L (L+1) + 8 + 8 L workgroups (248 at L = 6) of ~2-4 us each, four kernels. The radix-4 steps are done redundantly by
// every consumer (each needs all four quarters of its input anyway); that costs compute units, which idle otherwise.
// Everything in memory is natural order and read / written with a thread's words 256 apart (coalesced); the sub-transforms'
// "B" register order is reached through one extra LDS re-deal (a lane would otherwise own 16 adjacent words). Keys are a
// natural-order copy (hexl_ks_plan::d_keys_nat). Arithmetic, reduction schedules and bounds are those of the monolithic transforms
// (f64_arith.hpp): the same stage functions run with the same global stage numbers, so results are bit-identical.
#include <stdlib.h>

#include "hexl_internal.hpp"
#include "ntt_core_f64.hpp"

using namespace hx;

struct KsArgsQ {
    const KsModF64* mods;    // [K]
    const double* tables;    // [K][4][n]: w, w/p, inverse w (first entry at index 1), inverse w/p
    const double* keys;      // [L][L+1][2][n] centred, NATURAL order
    // per instance b = blockIdx.y (a handful at most):
    double* sub;             // [b][L][n]   the four sub-inverses of t_target[d], natural order inside each quarter
    double* subsp;           // [b][2][n]   the same for the accumulated special limb
    unsigned long long* prod;  // [b][2][L+1][n] integer accumulators, natural order
    const u64* t_target;     // [b][L][n]
    u64* result;             // [b][2][L][n]
    u32 L, K;
    u32* range_flag;
    u32 overwrite, skip;     // as in keyswitch_f64.hip
    // The host-pointer entry point runs a lone keyswitch ZERO-COPY (capi.hip keyswitch_host_lone, round 5): t_target and result are then
    // device-visible PINNED HOST memory -- k_ksq_intt pulls the 0.8 MB of t_target across PCIe while it transforms (18.7 us against a
    // 29.5 us hipMemcpyAsync in front of the kernel, tools/zero_copy_probe), k_ksq_down pushes its 1.6 MB the same way (31.6 against
    // 43.2 us behind it). Hence: t_target is read exactly ONCE (k_ksq_intt leaves the raw words in `tcopy` for the slot == d terms of
    // k_ksq_up), and k_ksq_down publishes every finished quarter limb in `done` so that the host can start adding limb by limb while
    // the rest is still in flight.
    u64* tcopy;              // [b][L][n] raw t_target words (device memory)
    u32* done;               // pinned host words [b][2][L][4], written with `epoch` behind the quarter's result words; or null
    u32* started;            // pinned host word: k_ksq_down's first workgroup writes `epoch` when it starts (the host then wakes its adders)
    u32* gate;               // device word, zeroed by k_ksq_intt: quarter limbs of k_ksq_down published so far, or null = no ordering
    u32 epoch;
    u32 flag_on_host;        // range_flag is pinned host memory: report with a plain system-scope store, not an atomic OR
    unsigned long long tiermap;   // kernels built with LAZY = -1 (plans of mixed tiers): nibble i = reduction period of limb i (with_tier)
};

// A sub-transform workgroup is alone on its compute unit: one wave per SIMD, nobody to cover a dependent FP64 chain's latency but
// the wave's own independent work. Tell the backend that occupancy is not a goal here (it otherwise schedules for few registers,
// i.e. one chain after the other).
#define KSQ_ILP __attribute__((amdgpu_waves_per_eu(1, 2)))
#ifndef KSQ_DEFAULT_MAX_PAIRS
#define KSQ_DEFAULT_MAX_PAIRS 128    // batches with at most this many (slot, d) pairs take the quarter-transform path by default
#endif

constexpr int QLOGN = 14, QLOGM = 12, QLOGE = 4;
using GQ = Geom<QLOGM, QLOGE>;                                   // 256 threads x 16 coefficients, three full passes
constexpr int QM = 1 << QLOGM;

// twelve stages of a 2^14-point transform on quarter `q`: stage numbers and twiddle group indices are the FULL transform's
// (forward: global stages 3..14, group index = q on top of the local one; inverse: global stages 1..12)
template <int LAZY>
struct WgSubNtt {
    static constexpr int E = GQ::E, P = GQ::P;
    static_assert(GQ::KL == QLOGE && GQ::NG == 1, "three full passes");

    // forward: A order of the quarter in, B order out; |out| <= 2.14p (no reduction after the last stage), as the monolithic FINAL = false
    template <int PASS>
    __device__ static __forceinline__ void fwd_pass(double (&v)[E], double* lds, int tid, u32 q, const double* w, const Mod m) {
        constexpr int S0L = PASS * QLOGE + 1;                    // first local stage of this pass (1-based)
        if constexpr (PASS < P - 1) {
            constexpr int LO = QLOGM - (PASS + 1) * QLOGE;
            const u32 Gl = (PASS == 0) ? 0u : (LO >= 6 ? u32(__builtin_amdgcn_readfirstlane(u32(tid) >> LO)) : (u32(tid) >> LO));
            const u32 Gg = (q << (S0L - 1)) | Gl;
            fwd_stages_f64<E, 0, QLOGE, S0L + 2, 0, LAZY, (PASS == 0 || LO >= 6)>(v, Gg, w, w, m);
            redeal_pass<GQ, LO, QLOGE, true, true, (PASS + 1 == P - 1)>(v, lds, tid);
            fwd_pass<PASS + 1>(v, lds, tid, q, w, m);
        } else {
            const u32 Gg = (q << (S0L - 1)) | u32(GQ::grpB(0, tid));
            fwd_stages_f64<E, 0, QLOGE, S0L + 2, 0, LAZY, false>(v, Gg, w, w, m);
        }
    }
    // inverse: B order of the quarter in, A order out, centred-ish (|x| <= ~1.9p product outputs, see f64_arith.hpp), NOT scaled
    template <int PASS>
    __device__ static __forceinline__ void inv_pass(double (&v)[E], double* lds, int tid, u32 q, const double* iw, const Mod m) {
        const hxf::InvScale none{0, 0, 0, 0};
        if constexpr (PASS == 0) {
            // coefficient index >> (LO + K) with LO = 0, K = 4: the quarter on top of the lane's group
            const u32 Gg = (q << (QLOGM - QLOGE)) | u32(GQ::grpB(0, tid));
            inv_stages_f64<E, 0, QLOGE, 0, QLOGN, false, LAZY, false, true>(v, Gg, iw, iw, m, none);
            inv_pass<1>(v, lds, tid, q, iw, m);
        } else if constexpr (PASS < P) {
            constexpr int LO = PASS * QLOGE;
            redeal_pass<GQ, LO, QLOGE, false, true, (PASS == 1)>(v, lds, tid);
            const u32 Gl = LO + QLOGE >= QLOGM ? 0u : (LO >= 6 ? u32(__builtin_amdgcn_readfirstlane(u32(tid) >> LO)) : (u32(tid) >> LO));
            const u32 Gg = (q << (QLOGM - LO - QLOGE)) | Gl;
            inv_stages_f64<E, 0, QLOGE, LO, QLOGN, false, LAZY, (LO + QLOGE >= QLOGM || LO >= 6), true>(v, Gg, iw, iw, m, none);
            inv_pass<PASS + 1>(v, lds, tid, q, iw, m);
        }
    }
};

// A order (word r * 256 + tid of the quarter) <-> B order through LDS
__device__ __forceinline__ void a_to_b(double (&v)[GQ::E], double* lds, int tid) {
    redeal_x<GQ, false, true>(v, lds, tid, [](int r, int t) { return GQ::idxA(r, t); }, [](int r, int t) { return GQ::idxB(r, t); });
}
__device__ __forceinline__ void b_to_a(double (&v)[GQ::E], double* lds, int tid) {
    redeal_x<GQ, false, true>(v, lds, tid, [](int r, int t) { return GQ::idxB(r, t); }, [](int r, int t) { return GQ::idxA(r, t); });
}

// The four quarters' sub-inverse outputs at one local index -> the four coefficients c[j + k n/4] of the finished inverse:
// global inverse stages 13 and 14 (the latter with n^-1 folded in) on registers a[k] = quarter k, exactly as the monolithic
// transform's last pass does them. Outputs centred (|x| <= p/2 + 2).
template <int LAZY>
__device__ __forceinline__ void inverse_finish(double (&a)[4], const double* iw, const Mod m, const hxf::InvScale sc) {
    inv_stages_f64<4, 0, 2, QLOGM, QLOGN, true, LAZY, true, true>(a, 0u, iw, iw, m, sc);
}
// global forward stages 1 and 2 on the four quarters (inputs centred, |x| <= 0.625p): a[k] becomes quarter k's input to its
// twelve remaining stages
template <int LAZY>
__device__ __forceinline__ void forward_start(double (&a)[4], const double* w, const Mod m) {
    fwd_stages_f64<4, 0, 2, 1, 0, LAZY, true>(a, 0u, w, w, m);
}

// the input-range flag of a launch whose flag word may be pinned host memory: every reporter stores the same 1 (no read-modify-write
// across PCIe, no dependence on PCIe AtomicOps)
__device__ __forceinline__ void report_range_q(hxf::RangeMask bad, const KsArgsQ& a) {
    if (a.flag_on_host) {
        if (bad != 0 && (threadIdx.x & 63) == 0) __hip_atomic_store(a.range_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        hxf::report_range(bad, a.range_flag);
    }
}

__device__ __forceinline__ u64 fold_below_q4(u64 x, u64 q) {     // x < 16 q -> x mod q
#pragma unroll
    for (int s = 3; s >= 0; --s) { const u64 mq = q << s; x = x >= mq ? x - mq : x; }
    return x;
}

// ---- K1: (d, q) ----------------------------------------------------------------------------------------------------------
template <int LAZY>
__global__ __launch_bounds__(GQ::T) KSQ_ILP void k_ksq_intt(KsArgsQ a) {
    extern __shared__ __attribute__((aligned(16))) double ldsq[];
    const int tid = threadIdx.x;
    const u32 d = blockIdx.x >> 2, q = blockIdx.x & 3, L = a.L, b = blockIdx.y;
    // zero this workgroup's share of the instance's integer accumulators: 2 (L+1) rows of n words, dealt over the 4 L workgroups by quarter rows
    unsigned long long* const pb = a.prod + size_t(b) * 2 * (L + 1) * (1 << QLOGN);
    for (u32 row = blockIdx.x; row < 2 * (L + 1) * 4; row += 4 * L)
#pragma unroll
        for (int r = 0; r < GQ::E; ++r) (pb + size_t(row) * QM + GQ::idxA(r, 0))[u32(tid)] = 0;
    if (a.gate && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.gate = 0;   // (k_ksq_down, three kernels later, counts in it)
    const KsModF64 md = a.mods[d];
    const u64 qd = (u64)md.m.p;
    const u64* src = a.t_target + (size_t(b) * L + d) * (1 << QLOGN) + size_t(q) * QM;
    double v[GQ::E];
    hxf::RangeMask bad = 0;
    u64* keep = a.tcopy + (size_t(b) * L + d) * (1 << QLOGN) + size_t(q) * QM;
    u64 raw[GQ::E];
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) raw[r] = (src + GQ::idxA(r, 0))[u32(tid)];           // the only read of t_target (it may lie across PCIe)
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) {
        (keep + GQ::idxA(r, 0))[u32(tid)] = raw[r];
        v[r] = hxf::to_f64_lt52_checked(raw[r], qd, bad);                                 // canonical, as they are
    }
    report_range_q(bad, a);
    a_to_b(v, ldsq, tid);
    const double* tb = a.tables + size_t(d) * 4 * (1 << QLOGN);
    with_tier<LAZY, false>(a.tiermap, d, [&](auto T) {
        WgSubNtt<decltype(T)::value>::template inv_pass<0>(v, ldsq, tid, q, tb + 2 * (1 << QLOGN), md.m);
    });
    double* dst = a.sub + (size_t(b) * L + d) * (1 << QLOGN) + size_t(q) * QM;
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) (dst + GQ::idxA(r, 0))[u32(tid)] = v[r];
}

// ---- K2: (slot, d, q) ------------------------------------------------------------------------------------------------------
template <int LAZY>
__global__ __launch_bounds__(GQ::T) KSQ_ILP void k_ksq_up(KsArgsQ a) {
    extern __shared__ __attribute__((aligned(16))) double ldsq[];
    const int tid = threadIdx.x;
    const u32 L = a.L, b = blockIdx.y;
    const u32 q = blockIdx.x & 3, sd = blockIdx.x >> 2;
    const u32 slot = sd / L, d = sd - slot * L;
    const u32 i = slot < L ? slot : a.K - 1;
    const KsModF64 mi = a.mods[i];
    const Mod m = mi.m;
    constexpr size_t N = size_t(1) << QLOGN;
    const double* k0 = a.keys + (size_t(d) * (L + 1) + slot) * 2 * N + size_t(q) * QM;      // key[d][slot][0], quarter q; [1] is N further
    double v[GQ::E];
    // the key words come from HBM (the 14.7 MB key set is cold for a lone keyswitch): requested as early as possible, but BEHIND the
    // sub-result loads the radix-4 loop waits for (vector memory returns in order)
    double ka[GQ::E], kb[GQ::E];
    auto request_keys = [&] {
#pragma unroll
        for (int r = 0; r < GQ::E; ++r) { ka[r] = (k0 + GQ::idxA(r, 0))[u32(tid)]; kb[r] = (k0 + N + GQ::idxA(r, 0))[u32(tid)]; }
    };
    if (slot == d) {                                              // NTT(INTT(t_d) mod q_d) = t_d
        const u64* src = a.tcopy + (size_t(b) * L + d) * N + size_t(q) * QM;    // k_ksq_intt's copy of t_target[d]
        u64 raw[GQ::E];
#pragma unroll
        for (int r = 0; r < GQ::E; ++r) raw[r] = (src + GQ::idxA(r, 0))[u32(tid)];
        request_keys();
#pragma unroll
        for (int r = 0; r < GQ::E; ++r) v[r] = hxf::reduce(hxf::to_f64(raw[r]), m);
    } else {
        const KsModF64 md = a.mods[d];
        const double* sb = a.sub + (size_t(b) * L + d) * N;
        const double* tbd = a.tables + size_t(d) * 4 * N;
        const double* tbi = a.tables + size_t(i) * 4 * N;
        double sv[GQ::E][4];
#pragma unroll
        for (int r = 0; r < GQ::E; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[r][k] = (sb + size_t(k) * QM + GQ::idxA(r, 0))[u32(tid)];
        request_keys();
        // the inverse's last two stages run modulo q_d in q_d's tier (strict or lazy: the two forms an inverse butterfly has), the forward
        // transform modulo q_i in q_i's
        with_tier<LAZY, false>(a.tiermap, d, [&](auto TD) {
#pragma unroll
            for (int r = 0; r < GQ::E; ++r) {
                inverse_finish<decltype(TD)::value>(sv[r], tbd + 2 * N, md.m, md.sc);      // c_d at j, j + n/4, j + n/2, j + 3n/4
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[r][k] = hxf::reduce(hxf::lift(sv[r][k], md.m), m);   // canonical, then mod q_i (intt1_redu.hpp:36-42)
            }
        });
        with_tier<LAZY, true>(a.tiermap, i, [&](auto TI) {
#pragma unroll
            for (int r = 0; r < GQ::E; ++r) {
                forward_start<decltype(TI)::value>(sv[r], tbi, m);
                v[r] = q == 0 ? sv[r][0] : q == 1 ? sv[r][1] : q == 2 ? sv[r][2] : sv[r][3];
            }
            WgSubNtt<decltype(TI)::value>::template fwd_pass<0>(v, ldsq, tid, q, tbi, m);
        });
        b_to_a(v, ldsq, tid);
    }
    unsigned long long* p0 = a.prod + (size_t(b) * 2 * (L + 1) + slot) * N + size_t(q) * QM;
    unsigned long long* p1 = p0 + size_t(L + 1) * N;
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) {
        const u64 t0 = hxf::from_f64(hxf::lift(hxf::reduce(hxf::mul_mod(v[r], ka[r], m), m), m));
        const u64 t1 = hxf::from_f64(hxf::lift(hxf::reduce(hxf::mul_mod(v[r], kb[r], m), m), m));
        __hip_atomic_fetch_add(p0 + GQ::idxA(r, 0) + tid, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(p1 + GQ::idxA(r, 0) + tid, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- K3a: (k, q) -----------------------------------------------------------------------------------------------------------
template <int LAZY>
__global__ __launch_bounds__(GQ::T) KSQ_ILP void k_ksq_intt_sp(KsArgsQ a) {
    extern __shared__ __attribute__((aligned(16))) double ldsq[];
    const int tid = threadIdx.x;
    const u32 k = blockIdx.x >> 2, q = blockIdx.x & 3, L = a.L, b = blockIdx.y;
    constexpr size_t N = size_t(1) << QLOGN;
    const KsModF64 msp = a.mods[a.K - 1];
    const u64 qsp = (u64)msp.m.p;
    const unsigned long long* src = a.prod + (size_t(b) * 2 * (L + 1) + k * (L + 1) + L) * N + size_t(q) * QM;
    double v[GQ::E];
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) v[r] = hxf::to_f64_lt52(fold_below_q4((src + GQ::idxA(r, 0))[u32(tid)], qsp));
    a_to_b(v, ldsq, tid);
    const double* ts = a.tables + size_t(a.K - 1) * 4 * N;
    with_tier<LAZY, false>(a.tiermap, a.K - 1, [&](auto T) {
        WgSubNtt<decltype(T)::value>::template inv_pass<0>(v, ldsq, tid, q, ts + 2 * N, msp.m);
    });
    double* dst = a.subsp + (size_t(b) * 2 + k) * N + size_t(q) * QM;
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) (dst + GQ::idxA(r, 0))[u32(tid)] = v[r];
}

// ---- K3b: (k, i, q) --------------------------------------------------------------------------------------------------------
template <int LAZY>
__global__ __launch_bounds__(GQ::T) KSQ_ILP void k_ksq_down(KsArgsQ a) {
    extern __shared__ __attribute__((aligned(16))) double ldsq[];
    const int tid = threadIdx.x;
    const u32 L = a.L, b = blockIdx.y;
    const u32 q = blockIdx.x & 3, ki = blockIdx.x >> 2;
    const u32 k = ki / L, i = ki - k * L;
    constexpr size_t N = size_t(1) << QLOGN;
    const KsModF64 msp = a.mods[a.K - 1], md = a.mods[i];
    const Mod m = md.m;
    const double* sb = a.subsp + (size_t(b) * 2 + k) * N;
    const double* ts = a.tables + size_t(a.K - 1) * 4 * N;
    const double* tb = a.tables + size_t(i) * 4 * N;
    const unsigned long long* pi = a.prod + (size_t(b) * 2 * (L + 1) + k * (L + 1) + i) * N + size_t(q) * QM;
    u64* res = a.result + ((size_t(b) * 2 + k) * L + i) * N + size_t(q) * QM;
    if (a.done && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        __hip_atomic_store(a.started, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    u64 praw[GQ::E], old[GQ::E];
    double v[GQ::E];
    double sv[GQ::E][4];
#pragma unroll
    for (int r = 0; r < GQ::E; ++r)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) sv[r][kk] = (sb + size_t(kk) * QM + GQ::idxA(r, 0))[u32(tid)];
    // the epilogue's inputs behind them (in-order return): they land while the transform runs
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) praw[r] = (pi + GQ::idxA(r, 0))[u32(tid)];
    if (!a.overwrite) {
#pragma unroll
        for (int r = 0; r < GQ::E; ++r) old[r] = (res + GQ::idxA(r, 0))[u32(tid)];
    }
    with_tier<LAZY, false>(a.tiermap, a.K - 1, [&](auto TS) {          // the special prime's tier, then limb i's
#pragma unroll
        for (int r = 0; r < GQ::E; ++r) {
            inverse_finish<decltype(TS)::value>(sv[r], ts + 2 * N, msp.m, msp.sc);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // y = s' - floor(q_sp/2): the exact centred remainder (keyswitch_x.hip ksx_special_down; intt2_redu.hpp:25-51)
                const double cc = hxf::lift(sv[r][kk], msp.m);
                sv[r][kk] = cc > msp.half ? cc - msp.m.p : cc;
                if (!a.skip) sv[r][kk] = hxf::reduce(sv[r][kk], m);
            }
        }
    });
    with_tier<LAZY, true>(a.tiermap, i, [&](auto TI) {
#pragma unroll
        for (int r = 0; r < GQ::E; ++r) {
            forward_start<decltype(TI)::value>(sv[r], tb, m);
            v[r] = q == 0 ? sv[r][0] : q == 1 ? sv[r][1] : q == 2 ? sv[r][2] : sv[r][3];
        }
        WgSubNtt<decltype(TI)::value>::template fwd_pass<0>(v, ldsq, tid, q, tb, m);   // |w| <= 2.14p
    });
    b_to_a(v, ldsq, tid);
    const u64 qi = (u64)m.p;
    hxf::RangeMask bad = 0;
    u64 outw[GQ::E];
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) {
        const double pv = hxf::reduce(hxf::to_f64_lt52(fold_below_q4(praw[r], qi)), m);
        const double out = hxf::mul_shoup(pv - v[r], md.msf, md.msf_p, m);                 // ms.hpp:70-82
        double rr;
        if (a.overwrite) rr = hxf::reduce(out, m);                                         // host-pointer path: the HOST adds
        else rr = hxf::reduce(hxf::to_f64_lt52_checked(old[r], qi, bad) + out, m);         // fpga.cpp:453-457
        outw[r] = hxf::from_f64(hxf::lift(rr, m));
    }
    // Zero-copy output (a.done): all 8 L workgroups finish their arithmetic at about the same time, and left alone their 1.6 MB of
    // stores share the PCIe link evenly -- every limb would land at the very end. The gate lets the limbs through IN ORDER, about three
    // at a time (enough bytes in flight to fill the link), so the host adds limb x into the caller's array while limbs x + 1 ... are
    // still crossing. A workgroup only ever waits for lower-numbered ones, which were dispatched before it (the launcher only passes a
    // gate for grids that are resident at once anyway).
    const u32 limb = (b * 2 + k) * L + i;
    if (a.gate && limb >= 3) {
        if (tid == 0)
            // (bounded -- ADVICE r05: the gate only orders PCIe stores for latency, so a workgroup that has waited ~2 ms for lower-numbered
            // ones that are, against all expectation, not resident goes ahead ungated instead of spinning for ever)
            for (u32 polls = 0; __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4 * (limb - 2) && polls < (1u << 15); ++polls)
                __builtin_amdgcn_s_sleep(8);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < GQ::E; ++r) (res + GQ::idxA(r, 0))[u32(tid)] = outw[r];
    report_range_q(bad, a);
    if (a.done) {
        // this quarter of result[k][i] is complete: every thread's stores are released to the system scope, then ONE word says so
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(a.done + size_t(limb) * 4 + q, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (a.gate) __hip_atomic_fetch_add(a.gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int LAZY>
static int run_lat(hexl_ks_plan* p, const KsArgsQ& a, u32 nb) {
    static PerDeviceOnce once;
    constexpr size_t lds = GQ::LDS_USED;
    if (int rc0 = once.run(p->ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ksq_intt<LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ksq_up<LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ksq_intt_sp<LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ksq_down<LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            return 0;
        }))
        return rc0;
    hipStream_t st = p->cur;
    const u32 L = a.L;
    hipLaunchKernelGGL((k_ksq_intt<LAZY>), dim3(4 * L, nb), dim3(GQ::T), lds, st, a);
    hipLaunchKernelGGL((k_ksq_up<LAZY>), dim3(4 * L * (L + 1), nb), dim3(GQ::T), lds, st, a);
    hipLaunchKernelGGL((k_ksq_intt_sp<LAZY>), dim3(8, nb), dim3(GQ::T), lds, st, a);
    hipLaunchKernelGGL((k_ksq_down<LAZY>), dim3(8 * L, nb), dim3(GQ::T), lds, st, a);
    return (int)hipGetLastError();
}

// a few keyswitches (N = 16384, FP64 plan with natural-order keys) on p->cur; scratch = the plan's current lane (nb <= p->cap).
// HEXL_KS_LAT: 0 = off, 1 = the three-kernel path of keyswitch_f64.hip instead, 2 = this path for EVERY batch that fits a scratch
// chunk (tests), HEXL_KS_LAT_MAX = largest batch that takes it by default.
bool hx_ks_lat_applies(const hexl_ks_plan* p, size_t nb) {
    static const int lat = [] { const char* e = getenv("HEXL_KS_LAT"); return e ? atoi(e) : -1; }();
    static const long most = [] { const char* e = getenv("HEXL_KS_LAT_MAX"); return e ? atol(e) : -1L; }();
    // default: as long as the (slot, d) pairs of the batch are at most KSQ_DEFAULT_MAX_PAIRS -- measured (tools/batch_sweep.py, us per
    // launch, this path / the five kernels): L = 6: 49.4 / 71.4 at one keyswitch, 59.7 / 73.8 at two, 67.2 / 76.5 at three, 80.2 / 78.5
    // at four; L = 7: 54.6 / 73.7, 67.6 / 76.3, 79.9 / 78.4 at three
    // (and at most eight instances: the rule was measured at L = 6 and 7; short key chains would otherwise send dozens of instances here)
    const bool small = most >= 0 ? nb <= (size_t)most : (nb <= 8 && nb * p->L * (p->L + 1) <= size_t(KSQ_DEFAULT_MAX_PAIRS));
    return p->use_f64 && p->logn == QLOGN && p->d_keys_nat && p->L <= 15 && lat != 0 && lat != 1 && (small || lat == 2);
}
int hx_launch_keyswitch_lat(hexl_ks_plan* p, u64* d_result, const u64* d_t_target, size_t nb) {
    const size_t n = p->n, L = p->L;
    KsArgsQ a;
    a.mods = p->d_mods_f64; a.tables = p->d_tables_f64; a.keys = p->d_keys_nat;
    // scratch of the (b, d)-major pipeline, one instance: c [L n] | u ... | prod [2 (L+1) n] | s [2 n]  (hx_ks_f64_scratch_words)
    double* base = (double*)p->cur_scratch;
    a.sub = base;
    double* u = base + p->cap * L * n;
    double* prod = u + p->cap * (L + 1) * L * n;
    a.prod = reinterpret_cast<unsigned long long*>(prod);
    a.subsp = prod + p->cap * 2 * (L + 1) * n;
    a.tcopy = reinterpret_cast<u64*>(u);                          // (the u region of the (b, d)-major layout is free on this path)
    a.done = p->host_done; a.epoch = p->host_epoch;
    a.started = p->host_done ? p->host_flag + 1 : nullptr;
    // ordered output only while the whole grid of k_ksq_down (256 threads per workgroup: at least one per CU) is resident at once
    a.gate = (p->host_done && nb * 8 * L <= (size_t)p->ctx->num_cu) ? p->d_flag + 2 : nullptr;
    a.flag_on_host = p->host_flag ? 1u : 0u;
    a.t_target = d_t_target; a.result = d_result;
    a.L = (u32)L; a.K = p->K;
    a.range_flag = p->host_flag ? p->host_flag : p->d_flag;
    a.overwrite = p->overwrite_result ? 1u : 0u;
    a.skip = p->x_skip ? 1u : 0u;
    a.tiermap = 0;
    for (u32 i = 0; i < p->K; ++i) a.tiermap |= (unsigned long long)(p->tier[i] & 15u) << (4 * i);
    // limbs of different tiers: looked up per transform (with_tier) for a LONE keyswitch (bridge-seal's chain: 48.6 against 49.5 us); two to
    // four instances measured 2-10 % slower that way than on the plan-wide tier (tools/seal_chain_rate.py, round 5) and keep the latter --
    // HEXL_KS_PER_LIMB=2: the lookup for every batch (tests; keyswitch_f64.hip has the same knob)
    static const bool lookup = [] { const char* e = getenv("HEXL_KS_PER_LIMB"); return e && atoi(e) == 2; }();
    if (p->mixed && (nb == 1 || lookup)) return run_lat<-1>(p, a, (u32)nb);
    switch (p->f64_lazy) {
        case 12: return run_lat<12>(p, a, (u32)nb);
        case 6:  return run_lat<6>(p, a, (u32)nb);
        case 3:  return run_lat<3>(p, a, (u32)nb);
        default: return run_lat<0>(p, a, (u32)nb);
    }
}
