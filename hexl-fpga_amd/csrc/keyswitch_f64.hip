// keyswitch_f64.hip -- K4 on the FP64 pipe: the production keyswitch path when every modulus is < 2^52
// (the reference's own limit, host/src/keyswitch.cpp:32). Dataflow of SURVEY 2.1-K4 steps 1-7 as four kernels per
// chunk of instances -- k_ksf_up (steps 1-2), k_ksf_mac (3), k_ksf_intt_sp (4), k_ksf_moddown (5-7); small batches
// split steps 1-2 into k_ksf_intt + k_ksf_ntt_up. Arithmetic from f64_arith.hpp (exact integers in doubles), so
// results are bit-identical to the integer path (keyswitch.hip) and to the reference's canonical pipeline.
//
// HBM-resident intermediates are doubles: c (canonical, natural order; small batches only), u and prod in the
// forward transform's register order ("B order": stored as [register][thread], fully coalesced, and consumed the
// same way), s' (canonical, natural order); keys in B order too, twiddles centred. Only t_target (read) and result
// (read-modify-write) are converted from/to uint64.
#include <stdlib.h>

#include "hexl_internal.hpp"
#ifndef KSF_BIG_PRIO
#define KSF_BIG_PRIO 1222  // ... in k_ksf_ntt_up / k_ksf_moddown at N = 32768 (32 coefficients per thread, half-size re-deals)
#endif
#ifndef KSF_UP_PRIO
#define KSF_UP_PRIO 1222   // wave priority by pass in k_ksf_up (0 = off)
#endif
#include "ntt_core_f64.hpp"

using namespace hx;

struct KsArgsF {
    const KsModF64* mods;    // [K]
    const double* tables;    // [K][4][n]: w, w/p, inverse w (first entry at index 1), inverse w/p
    const double* keys;      // [L][L+1][2][n] centred, B order
    double* c;               // [chunk][L][n]          canonical, natural order
    double* u;               // [chunk][L+1][L][n]     |u| <= 2.14p (no final range reduction), B order
    double* prod;            // [chunk][2][L+1][n]     centred, B order
    double* s;               // [chunk][2][n]          canonical, natural order
    const u64* t_target;     // [chunk][L][n]
    u64* result;             // [chunk][2][L][n]
    u32 L, K, nb;
    u32* range_flag;         // set to 1 when a t_target / result word is not below its modulus (hexl_ks_range_check)
    u32 overwrite;           // 1: `result` is written, not accumulated into (the host-pointer path: the HOST adds, fpga.cpp:441-475)
    u32 skip;                // latency path: moduli of one size (hexl_ks_plan::x_skip) -> s' enters the mod-down transform un-reduced
    unsigned long long tiermap;   // kernels built with LAZY = -1 (plans of mixed tiers): nibble i = reduction period of limb i
};

__device__ __forceinline__ u32 xcd_item_f(u32 bid, u32 total) {   // see keyswitch.hip: XCD-contiguous work ranges
    const u32 q = total >> 3, r = total & 7, xcd = bid & 7, j = bid >> 3;
    return xcd * q + (xcd < r ? xcd : r) + j;
}

// ---- small batches: one transform per workgroup, so that even a single keyswitch spreads over L*L + ... CUs ----
// step 1: c_d = INTT_{q_d}(t_target[d]) as canonical doubles
// mixed-tier kernels (LAZY = -1): four schedules per forward transform at N = 16384 (two -- lazy period 3 / strict -- measured no faster)
template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksf_intt(KsArgsF a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const u32 item = blockIdx.x;                                  // b*L + d
    const u32 d = __builtin_amdgcn_readfirstlane(item % a.L);
    const KsModF64 md = a.mods[d];
    const double* tb = a.tables + size_t(d) * 4 * G::N;
    const u64* src = a.t_target + size_t(item) * G::N;
    double v[G::E];
    hxf::RangeMask bad = 0;
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(hxf::to_f64_checked(src[G::idxB(r, tid)], md.m, bad), md.m);
    hxf::report_range(bad, a.range_flag);
    // step 2 for slot == d needs no transform: NTT_{q_d}(INTT_{q_d}(t_d) mod q_d) = t_d (the reference recomputes
    // it; same value for in-range data). The registers already hold t_d in B order.
    {
        const u32 b = item / a.L;
        double* ud = a.u + ((size_t(b) * (a.L + 1) + d) * a.L + d) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) ud[r * G::T + tid] = v[r];
    }
    with_tier<LAZY, false>(a.tiermap, d, [&](auto T) {             // (an inverse transform has two forms: strict and lazy)
        WgNttF64<LOGN, LOGE, decltype(T)::value>::template inverse<true>(v, ldsd, tid, tb + 2 * G::N, tb + 3 * G::N, md.m, md.sc);
    });
    double* dst = a.c + size_t(item) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r) dst[G::idxA(r, tid)] = hxf::lift(v[r], md.m);
}

// step 2: u[b][slot][d] = NTT_{q_i}(c_d mod q_i) for slot != d (slot == d is written by k_ksf_intt), one
// transform per workgroup, kept in the forward transform's register order ("B order", fully coalesced).
// (Capping this kernel at 96 VGPRs so that a k_ksf_mac workgroup of the other lane could be co-resident was
// measured: +12 % instructions, no throughput gain -- not done.)
template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksf_ntt_up(KsArgsF a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const u32 L = a.L;
    // (b*L + d)*L + s, XCD-contiguous: the L transforms that read the same c_d run back to back on one XCD, so c_d
    // comes from HBM once and from that XCD's L2 afterwards
    const u32 item = __builtin_amdgcn_readfirstlane(xcd_item_f(blockIdx.x, gridDim.x));
    const u32 bd = item / L, sidx = item - bd * L;
    const u32 b = bd / L, d = bd - b * L;
    const u32 slot = sidx + (sidx >= d ? 1u : 0u);                // 0..L without d; slot L is the special prime
    const u32 i = slot < L ? slot : a.K - 1;
    const KsModF64 md = a.mods[i];
    const Mod m = md.m;
    double* dst = a.u + ((size_t(b) * (L + 1) + slot) * L + d) * G::N;
    double v[G::E];
    const double* cd = a.c + (size_t(b) * L + d) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(cd[G::idxA(r, tid)], m);          // c_d mod q_i (intt1_redu.hpp:36-42)
    const double* tb = a.tables + size_t(i) * 4 * G::N;
    // no final range reduction (LAZY): |u| <= 2.14p, which mul_mod in k_ksf_mac accepts (|u.key| < 2^102,
    // |result| < p); tests/cpp/f64_selftest.cpp replays exactly this chain against 128-bit integers
    with_tier<LAZY, LOGN == 14>(a.tiermap, i, [&](auto T) {
        using W = WgNttF64<LOGN, LOGE, decltype(T)::value, 0, 0, false, (LOGN >= 15 ? KSF_BIG_PRIO : 0)>;   // N = 32768: +7 % (batch 32 at N = 16384: -5 %)
        W::template forward<true, false>(v, ldsd, tid, tb, tb + G::N, m);
    });
#pragma unroll
    for (int r = 0; r < G::E; ++r) dst[r * G::T + tid] = v[r];
}

// ---- large batches: one workgroup per polynomial of the INPUT, all its transforms back to back ----
// steps 1-2 in one kernel: c_d = INTT_{q_d}(t_target[d]) (canonical), then u[b][slot][d] = NTT_{q_i}(c_d mod q_i)
// for every slot. One workgroup per (b, d) keeps c_d in registers (A order is both the inverse transform's output
// and the forward transform's input order) and runs the L forward transforms back to back: c never travels to
// memory, and only the first of the L+1 transforms waits for an input -- a lone workgroup per CU otherwise idles
// ~5 us per transform on that wait (tools/load_probe.hip, tools/ntt_timeline.hip). The waves of the workgroup only
// meet at each transform's cross-wave re-deal, so early waves start the next slot while late ones still store.
// u is kept in the forward transform's register order ("B order", fully coalesced).
template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksf_up(KsArgsF a) {
    static_assert(LAZY >= 0, "one schedule for all L + 1 transforms of the workgroup: plans of mixed tiers run k_ksf_intt + k_ksf_ntt_up");
    using G = Geom<LOGN, LOGE>;
    // FPRIO 1222: this workgroup runs L transforms back to back -- the pass in front of the cross-wave barrier at the lower wave
    // priority (ntt_core_f64.hpp hx_fwd_prio): N = 32768, L = 3, batch 2048: 143.5 k -> 163.5 k keyswitch/s (+14 %)
    using W = WgNttF64<LOGN, LOGE, LAZY, 0, 0, false, KSF_UP_PRIO>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const u32 L = a.L;
    const u32 item = blockIdx.x;                                  // b*L + d
    const u32 b = __builtin_amdgcn_readfirstlane(item / L), d = item - b * L;
    double c[G::E];
    {
        const int tid = threadIdx.x;
        const KsModF64 md = a.mods[d];
        const double* tb = a.tables + size_t(d) * 4 * G::N;
        const u64* src = a.t_target + size_t(item) * G::N;
        // uniform row pointer + unsigned 32-bit thread offset: SGPR-base addressing, no 64-bit VALU address math
        const u32 tB = u32(G::idxB(0, tid));
        hxf::RangeMask bad = 0;
#pragma unroll
        for (int r = 0; r < G::E; ++r) c[r] = hxf::reduce(hxf::to_f64_checked((src + G::idxB(r, 0))[tB], md.m, bad), md.m);
        hxf::report_range(bad, a.range_flag);
        // step 2 for slot == d needs no transform: NTT_{q_d}(INTT_{q_d}(t_d) mod q_d) = t_d (the reference
        // recomputes it; same value for in-range data). The registers already hold t_d in B order.
        double* ud = a.u + ((size_t(b) * (L + 1) + d) * L + d) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) __builtin_nontemporal_store(c[r], &(ud + r * G::T)[u32(tid)]);   // streamed to k_ksf_mac
        W::template inverse<true>(c, ldsd, tid, tb + 2 * G::N, tb + 3 * G::N, md.m, md.sc);
#pragma unroll
        for (int r = 0; r < G::E; ++r) c[r] = hxf::lift(c[r], md.m);                    // canonical c_d, A order
    }
#pragma unroll 1
    for (u32 s = 0; s < L; ++s) {
        const u32 slot = s + (s >= d ? 1u : 0u);                  // 0..L without d; slot L is the special prime
        const u32 i = slot < L ? slot : a.K - 1;
        const Mod m = a.mods[i].m;
        int tid = threadIdx.x;                                    // laundered per slot: otherwise every LDS / global
        asm volatile("" : "+v"(tid));                             // address is hoisted out of the loop and spilled
        u32 toff = i * 4 * G::N;                                  // laundered offset, not pointer: a laundered pointer
        asm volatile("" : "+s"(toff));                            // loses its address space and turns the loads into FLAT
        const double* tb = a.tables + toff;
        double v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(c[r], m);                     // c_d mod q_i (intt1_redu.hpp:36-42)
        // not FRESH: other waves may still be reading their LDS block of the previous transform. No final range
        // reduction (LAZY): |u| <= 2.14p, which mul_mod in k_ksf_mac accepts (tests/cpp/f64_selftest.cpp)
        W::template forward<false, false>(v, ldsd, tid, tb, tb + G::N, m);
        double* dst = a.u + ((size_t(b) * (L + 1) + slot) * L + d) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) __builtin_nontemporal_store(v[r], &(dst + r * G::T)[u32(tid)]);
    }
}

// step 3: prod[b][k][slot] = sum_d u[b][slot][d] . key[d][k][slot]  (dyadmult.hpp:128-140). Pure streaming:
// a thread owns two adjacent coefficients of one slot, keeps their 2*L*2 key words in registers and walks the
// batch, so keys are read once per launch and u / prod exactly once.
template <int MAXL>
__global__ __launch_bounds__(256) void k_ksf_mac(KsArgsF a, u32 n) {
    const u32 L = a.L;
    const u32 pairs = n >> 1;
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;        // (slot, pair)
    const u32 slot = gid / pairs;
    if (slot > L) return;
    const u32 j = (gid - slot * pairs) * 2;
    const u32 i = slot < L ? slot : a.K - 1;
    const Mod m = a.mods[i].m;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 key[MAXL][2];
#pragma unroll
    for (int d = 0; d < MAXL; ++d)
        if (d < (int)L) {
            key[d][0] = *reinterpret_cast<const d2*>(a.keys + ((size_t(d) * (L + 1) + slot) * 2 + 0) * n + j);
            key[d][1] = *reinterpret_cast<const d2*>(a.keys + ((size_t(d) * (L + 1) + slot) * 2 + 1) * n + j);
        }
    for (u32 b = blockIdx.y; b < a.nb; b += gridDim.y) {
        const double* ub = a.u + ((size_t(b) * (L + 1) + slot) * L) * n + j;
        d2 acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
#pragma unroll
        for (int d = 0; d < MAXL; ++d)
            if (d < (int)L) {
                const d2 u = __builtin_nontemporal_load(reinterpret_cast<const d2*>(ub + size_t(d) * n));   // read exactly once
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc0[e] = hxf::reduce(acc0[e] + hxf::mul_mod(u[e], key[d][0][e], m), m);
                    acc1[e] = hxf::reduce(acc1[e] + hxf::mul_mod(u[e], key[d][1][e], m), m);
                }
            }
        __builtin_nontemporal_store(acc0, reinterpret_cast<d2*>(a.prod + ((size_t(b) * 2 + 0) * (L + 1) + slot) * n + j));
        __builtin_nontemporal_store(acc1, reinterpret_cast<d2*>(a.prod + ((size_t(b) * 2 + 1) * (L + 1) + slot) * n + j));
    }
}

// step 4: s'_k = INTT_{q_sp}(prod[k][special]) + floor(q_sp/2)  (mod q_sp), canonical
template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksf_intt_sp(KsArgsF a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const u32 L = a.L;
    const u32 item = blockIdx.x;                                  // b*2 + k
    const u32 i = a.K - 1;
    const KsModF64 md = a.mods[i];
    const Mod m = md.m;
    const double* tb = a.tables + size_t(i) * 4 * G::N;
    const double* src = a.prod + (size_t(item) * (L + 1) + L) * G::N;
    double v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = __builtin_nontemporal_load(&src[r * G::T + tid]);
    with_tier<LAZY, false>(a.tiermap, i, [&](auto T) {
        WgNttF64<LOGN, LOGE, decltype(T)::value>::template inverse<true>(v, ldsd, tid, tb + 2 * G::N, tb + 3 * G::N, m, md.sc);
    });
    double* dst = a.s + size_t(item) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r)                                // intt2_redu.hpp:25,43
        dst[G::idxA(r, tid)] = hxf::lift(hxf::reduce(hxf::lift(v[r], m) + md.half, m), m);
}

// steps 5-7: w = NTT((s' + fix_i) mod q_i); result += (prod - w) * msf_i
template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksf_moddown(KsArgsF a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const u32 L = a.L;
    // (b*2 + k)*L + i, XCD-contiguous: the L transforms that read the same s'_k run back to back on one XCD
    const u32 item = __builtin_amdgcn_readfirstlane(xcd_item_f(blockIdx.x, gridDim.x));
    const u32 bk = item / L, i = item - bk * L;
    const u32 b = bk >> 1, k = bk & 1;
    const KsModF64 md = a.mods[i];
    const Mod m = md.m;
    const double* tb = a.tables + size_t(i) * 4 * G::N;

    const double* sk = a.s + (size_t(b) * 2 + k) * G::N;
    double v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(sk[G::idxA(r, tid)] + md.fix, m);   // intt2_redu.hpp:49-51
    const double* pk = a.prod + ((size_t(b) * 2 + k) * (L + 1) + i) * G::N;
    u64* res = a.result + ((size_t(b) * 2 + k) * L + i) * G::N;
    // prod is requested right after the cross-wave re-deal and lands during the remaining passes; result is
    // requested first thing in the epilogue and lands during the (prod - w) * msf multiplications
    double pv[G::E];
    with_tier<LAZY, LOGN == 14>(a.tiermap, i, [&](auto T) {
        using W = WgNttF64<LOGN, LOGE, decltype(T)::value, 0, 0, false, (LOGN >= 15 ? KSF_BIG_PRIO : 0)>;   // N = 32768: +7 % (N = 16384: +-0)
        if constexpr (G::HALF_ONLY) {                              // N = 32768: no registers to hold prod during the transform
            W::template forward<true, false>(v, ldsd, tid, tb, tb + G::N, m);
#pragma unroll
            for (int r = 0; r < G::E; ++r) pv[r] = (pk + r * G::T)[u32(tid)];
        } else {
            W::template forward<true, false>(v, ldsd, tid, tb, tb + G::N, m, [&] {    // |w| <= 2.14p: |prod - w| <= 2.64p below
#pragma unroll
                for (int r = 0; r < G::E; ++r) pv[r] = (pk + r * G::T)[u32(tid)];
            });
        }
    });
    const u32 tB = u32(G::idxB(0, tid));
    if (!G::HALF_ONLY && a.overwrite) {                           // host-pointer path: the output itself, canonical; the HOST adds
#pragma unroll
        for (int r = 0; r < G::E; ++r)
            (res + G::idxB(r, 0))[tB] = hxf::from_f64(hxf::lift(hxf::reduce(hxf::mul_shoup(pv[r] - v[r], md.msf, md.msf_p, m), m), m));
        return;
    }
    u64 old[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) old[r] = (res + G::idxB(r, 0))[tB];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::mul_shoup(pv[r] - v[r], md.msf, md.msf_p, m);    // ms.hpp:70-82
    hxf::RangeMask bad = 0;
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const double rr = hxf::reduce(hxf::to_f64_checked(old[r], m, bad) + v[r], m);            // fpga.cpp:453-457
        (res + G::idxB(r, 0))[tB] = hxf::from_f64(hxf::lift(rr, m));
    }
    hxf::report_range(bad, a.range_flag);
}

// ---- latency path (round 4): a LONE keyswitch -- the SEAL bridge's only call shape (experimental/bridge-seal/tests/
// fpga_context.h:13-16: set_worksize_KeySwitch(1)) -- in THREE dependent kernels instead of five. The dataflow is four transforms
// deep whatever the kernel count (INTT -> NTT -> INTT_sp -> NTT); what can go is kernel boundaries and memory round trips:
//   k_ksl_intt (b, d)        c_d = INTT(t_d) -> scratch; zeroes its share of the (integer) accumulator `prod`            step 1
//   k_ksl_up   (b, slot, d)  NTT_{q_slot}(c_d mod q_slot) (slot == d: t_d itself, no transform), times key[d][slot][k], canonical,
//                            ADDED into prod[k][slot] by 64-bit integer atomics: the sum over d of L <= 15 residues below 2^52
//                            stays below 2^56, is exact in any order, and costs no extra kernel                       steps 2-3
//   k_ksl_down (b, k, i)     every workgroup sums-down its own copy of the special limb: INTT_sp(prod[k][special]) stays in
//                            registers (the inverse's output order is the forward's input order), then NTT_{q_i}, then the
//                            mod-switch epilogue with prod[k][i] requested behind the cross-wave re-deal               steps 4-7
// L + L(L+1) + 2L workgroups of one transform each (54 + ... at L = 6): the chip is mostly idle, latency is all that counts.
// prod[.] mod q: the atomically accumulated word is below L q; four conditional subtractions bring it below q.
template <int MAXBITS = 4>
__device__ __forceinline__ u64 fold_below_q(u64 x, u64 q) {
#pragma unroll
    for (int s = MAXBITS - 1; s >= 0; --s) { const u64 m = q << s; x = x >= m ? x - m : x; }
    return x;
}

template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksl_intt(KsArgsF a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const u32 item = blockIdx.x;                                  // b*L + d
    const u32 L = a.L;
    const u32 b = item / L, d = __builtin_amdgcn_readfirstlane(item - b * L);
    // this workgroup's rows of the accumulator (2 (L+1) rows per instance, dealt round robin over its L workgroups)
    u64* acc = reinterpret_cast<u64*>(a.prod) + size_t(b) * 2 * (L + 1) * G::N;
    for (u32 row = d; row < 2 * (L + 1); row += L)
#pragma unroll
        for (int r = 0; r < G::E; ++r) (acc + size_t(row) * G::N + G::idxA(r, 0))[u32(tid)] = 0;
    const KsModF64 md = a.mods[d];
    const double* tb = a.tables + size_t(d) * 4 * G::N;
    const u64* src = a.t_target + size_t(item) * G::N;
    const u64 qd = (u64)md.m.p;
    double v[G::E];
    hxf::RangeMask bad = 0;
    const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::to_f64_lt52_checked((src + G::idxB(r, 0))[tB], qd, bad);   // canonical words as they are
    hxf::report_range(bad, a.range_flag);
    with_tier<LAZY, false>(a.tiermap, d, [&](auto T) {
        WgNttF64<LOGN, LOGE, decltype(T)::value, 0, 0, true>::template inverse<true>(v, ldsd, tid, tb + 2 * G::N, tb + 3 * G::N, md.m, md.sc);
    });
    double* dst = a.c + size_t(item) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r) (dst + G::idxA(r, 0))[u32(tid)] = hxf::lift(v[r], md.m);
}

template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksl_up(KsArgsF a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const u32 L = a.L;
    const u32 item = blockIdx.x;                                  // (b*(L+1) + slot)*L + d
    const u32 bs = item / L, d = item - bs * L;
    const u32 b = bs / (L + 1), slot = __builtin_amdgcn_readfirstlane(bs - b * (L + 1));
    const u32 i = slot < L ? slot : a.K - 1;
    const KsModF64 md = a.mods[i];
    const Mod m = md.m;
    const double* k0 = a.keys + (size_t(d) * (L + 1) + slot) * 2 * G::N;        // key[d][slot][0], [1] follows; B order
    double v[G::E], ka[G::E], kb[G::E];
    auto request_keys = [&] {
#pragma unroll
        for (int r = 0; r < G::E; ++r) { ka[r] = (k0 + r * G::T)[u32(tid)]; kb[r] = (k0 + G::N + r * G::T)[u32(tid)]; }
    };
    if (slot == d) {                                              // NTT(INTT(t_d) mod q_d) = t_d: no transform
        const u64* src = a.t_target + (size_t(b) * L + d) * G::N;
        const u32 tB = u32(G::idxB(0, tid));
        u64 raw[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (src + G::idxB(r, 0))[tB];
        request_keys();
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(hxf::to_f64(raw[r]), m);
    } else {
        const double* cd = a.c + (size_t(b) * L + d) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce((cd + G::idxA(r, 0))[u32(tid)], m);   // c_d mod q_i (intt1_redu.hpp:36-42)
        const double* tb = a.tables + size_t(i) * 4 * G::N;
        with_tier<LAZY, LOGN == 14>(a.tiermap, i, [&](auto T) {                                // |u| <= 2.14p; keys behind the cross-wave re-deal
            WgNttF64<LOGN, LOGE, decltype(T)::value>::template forward<true, false>(v, ldsd, tid, tb, tb + G::N, m, request_keys);
        });
    }
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(a.prod) + size_t(b) * 2 * (L + 1) * G::N;
    unsigned long long* p0 = acc + size_t(0 * (L + 1) + slot) * G::N;
    unsigned long long* p1 = acc + size_t(1 * (L + 1) + slot) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const u64 t0 = hxf::from_f64(hxf::lift(hxf::reduce(hxf::mul_mod(v[r], ka[r], m), m), m));
        const u64 t1 = hxf::from_f64(hxf::lift(hxf::reduce(hxf::mul_mod(v[r], kb[r], m), m), m));
        // relaxed, device scope, no return value: the order of the L additions does not matter for an integer sum
        __hip_atomic_fetch_add(p0 + r * G::T + tid, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(p1 + r * G::T + tid, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksl_down(KsArgsF a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x;
    const u32 L = a.L;
    const u32 item = blockIdx.x;                                  // (b*2 + k)*L + i
    const u32 bk = item / L, i = __builtin_amdgcn_readfirstlane(item - bk * L);
    const u32 b = bk >> 1, k = bk & 1;
    const KsModF64 msp = a.mods[a.K - 1], md = a.mods[i];
    const Mod m = md.m;
    const u64* acc = reinterpret_cast<const u64*>(a.prod) + size_t(b) * 2 * (L + 1) * G::N;
    const u64* psp = acc + size_t(k * (L + 1) + L) * G::N;        // accumulated special limb, B order
    const u64* pi = acc + size_t(k * (L + 1) + i) * G::N;
    double v[G::E];
    {
        const u64 qsp = (u64)msp.m.p;
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::to_f64_lt52(fold_below_q<4>((psp + r * G::T)[u32(tid)], qsp));
        const double* ts = a.tables + size_t(a.K - 1) * 4 * G::N;
        with_tier<LAZY, false>(a.tiermap, a.K - 1, [&](auto T) {
            WgNttF64<LOGN, LOGE, decltype(T)::value, 0, 0, true>::template inverse<true>(v, ldsd, tid, ts + 2 * G::N, ts + 3 * G::N, msp.m, msp.sc);
        });
        // y = s' - floor(q_sp/2), the exact centred remainder (keyswitch_x.hip ksx_special_down; intt2_redu.hpp:25-51), A order
#pragma unroll
        for (int r = 0; r < G::E; ++r) {
            const double c = hxf::lift(v[r], msp.m);
            v[r] = c > msp.half ? c - msp.m.p : c;
        }
    }
    if (!a.skip) {
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(v[r], m);
    }
    const double* tb = a.tables + size_t(i) * 4 * G::N;
    u64 praw[G::E];
    with_tier<LAZY, LOGN == 14>(a.tiermap, i, [&](auto T) {
        WgNttF64<LOGN, LOGE, decltype(T)::value>::template forward<false, false>(v, ldsd, tid, tb, tb + G::N, m, [&] {     // |w| <= 2.14p
#pragma unroll
            for (int r = 0; r < G::E; ++r) praw[r] = (pi + r * G::T)[u32(tid)];
        });
    });
    const u64 qi = (u64)m.p;
    u64* res = a.result + ((size_t(b) * 2 + k) * L + i) * G::N;
    const u32 tB = u32(G::idxB(0, tid));
    if (a.overwrite) {                                            // host-pointer path: the output itself, canonical; the HOST adds
#pragma unroll
        for (int r = 0; r < G::E; ++r) {
            const double pv = hxf::reduce(hxf::to_f64_lt52(fold_below_q<4>(praw[r], qi)), m);
            (res + G::idxB(r, 0))[tB] = hxf::from_f64(hxf::lift(hxf::reduce(hxf::mul_shoup(pv - v[r], md.msf, md.msf_p, m), m), m));
        }
        return;
    }
    u64 old[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) old[r] = (res + G::idxB(r, 0))[tB];
    hxf::RangeMask bad = 0;
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const double pv = hxf::reduce(hxf::to_f64_lt52(fold_below_q<4>(praw[r], qi)), m);
        const double out = hxf::mul_shoup(pv - v[r], md.msf, md.msf_p, m);                          // ms.hpp:70-82
        const double rr = hxf::reduce(hxf::to_f64_lt52_checked(old[r], qi, bad) + out, m);          // fpga.cpp:453-457
        (res + G::idxB(r, 0))[tB] = hxf::from_f64(hxf::lift(rr, m));
    }
    hxf::report_range(bad, a.range_flag);
}

// ---------------------------------------------------------------------------------------------
template <class K>
static int set_lds(K kern, size_t bytes) {
    HX_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

template <int LOGN, int LOGE, int LAZY>
static int run_chunk_f64(hexl_ks_plan* p, const KsArgsF& a, int stage_mask, hipEvent_t* ev) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc0 = once.run(p->ctx->device, [] {
            int rc = 0;
            if constexpr (LAZY >= 0) rc = set_lds(k_ksf_up<LOGN, LOGE, LAZY>, G::LDS_USED);
            if (!rc) rc = set_lds(k_ksf_intt<LOGN, LOGE, LAZY>, G::LDS_USED);
            if (!rc) rc = set_lds(k_ksf_ntt_up<LOGN, LOGE, LAZY>, G::LDS_USED);
            if (!rc) rc = set_lds(k_ksf_intt_sp<LOGN, LOGE, LAZY>, G::LDS_USED);
            if (!rc) rc = set_lds(k_ksf_moddown<LOGN, LOGE, LAZY>, G::LDS_USED);
            return rc;
        }))
        return rc0;
    hipStream_t st = p->cur;
    const u32 L = a.L, nb = a.nb;
    // latency path (three kernels, above): a LONE keyswitch. Measured (tools/batch_sweep.py, N = 16384, device-resident): 70.3 us
    // against 72.5 us for the five kernels at L = 6, 72.2 against 73.2 at L = 7 -- the four dependent transforms, ~14 us each for a
    // lone workgroup (9-10 us of FP64 issue on ONE CU + a cold 128 KiB load + a kernel boundary), are what bounds it, not the
    // kernel count; from two keyswitches up the five kernels win (their per-(slot, d) workgroups skip the redundant special-limb
    // inverse: 73.5 against 78.3 us at two, 78.4 against 96.9 at four). HEXL_KS_LAT=0 turns it off, 1 forces it for every batch
    // that takes this pipeline (tests).
    // Not for N = 32768 (no registers for the key rows beside 64 data registers); timing runs (ev) keep the five-kernel path,
    // whose stages the events bracket.
    static const int lat = [] { const char* e = getenv("HEXL_KS_LAT"); return e ? atoi(e) : -1; }();
    if constexpr (!G::HALF_ONLY)
    if (!ev && stage_mask == 7 && L <= 15 && (lat == 1 || (lat != 0 && nb == 1))) {
        static PerDeviceOnce once_l;
        if (int rc0 = once_l.run(p->ctx->device, [] {
                int rc = set_lds(k_ksl_intt<LOGN, LOGE, LAZY>, G::LDS_USED);
                if (!rc) rc = set_lds(k_ksl_up<LOGN, LOGE, LAZY>, G::LDS_USED);
                if (!rc) rc = set_lds(k_ksl_down<LOGN, LOGE, LAZY>, G::LDS_USED);
                return rc;
            }))
            return rc0;
        hipLaunchKernelGGL((k_ksl_intt<LOGN, LOGE, LAZY>), dim3(nb * L), dim3(G::T), G::LDS_USED, st, a);
        hipLaunchKernelGGL((k_ksl_up<LOGN, LOGE, LAZY>), dim3(nb * (L + 1) * L), dim3(G::T), G::LDS_USED, st, a);
        hipLaunchKernelGGL((k_ksl_down<LOGN, LOGE, LAZY>), dim3(nb * 2 * L), dim3(G::T), G::LDS_USED, st, a);
        return (int)hipGetLastError();
    }
    // one workgroup per input polynomial (all its transforms back to back) once that alone fills the chip twice;
    // below that one workgroup per transform, so that small batches still spread over the CUs
    const u32 cus = (u32)p->ctx->num_cu;
    // (the same fusion of steps 4-7 -- s' in registers, L mod-down transforms per workgroup -- measured 8 % slower
    // than the two kernels below: its epilogue loads cannot be requested early, tools/experiments/fused_down.patch)
    static const int fuse = [] { const char* e = getenv("HEXL_KS_FUSE"); return e ? atoi(e) : 1; }();
    const bool fused_up = LAZY >= 0 && (fuse & 1) && nb * L >= 2 * cus && !G::HALF_ONLY;   // N = 32768: 64 VGPRs of data already
    // timing stages: 1 = steps 1-2 (inverse + mod-up transforms), 2 = steps 3-4, 4 = steps 5-7
    if (ev) HX_CHECK(hipEventRecord(ev[0], st));
    if (stage_mask & 1) {
        if (fused_up) {
            if constexpr (LAZY >= 0) hipLaunchKernelGGL((k_ksf_up<LOGN, LOGE, LAZY>), dim3(nb * L), dim3(G::T), G::LDS_USED, st, a);
        } else {
            hipLaunchKernelGGL((k_ksf_intt<LOGN, LOGE, LAZY>), dim3(nb * L), dim3(G::T), G::LDS_USED, st, a);
            hipLaunchKernelGGL((k_ksf_ntt_up<LOGN, LOGE, LAZY>), dim3(nb * L * L), dim3(G::T), G::LDS_USED, st, a);
        }
    }
    if (ev) HX_CHECK(hipEventRecord(ev[1], st));
    if (stage_mask & 2) {
        const u32 threads = (L + 1) * (G::N / 2);
        const u32 by = nb < 8 ? nb : 8;                            // 8 batch lanes keep >= 2048 workgroups in flight
        if (L <= 8) hipLaunchKernelGGL((k_ksf_mac<8>), dim3(threads / 256, by), dim3(256), 0, st, a, (u32)G::N);
        else        hipLaunchKernelGGL((k_ksf_mac<16>), dim3(threads / 256, by), dim3(256), 0, st, a, (u32)G::N);
        hipLaunchKernelGGL((k_ksf_intt_sp<LOGN, LOGE, LAZY>), dim3(nb * 2), dim3(G::T), G::LDS_USED, st, a);
    }
    if (ev) HX_CHECK(hipEventRecord(ev[2], st));
    if (stage_mask & 4)
        hipLaunchKernelGGL((k_ksf_moddown<LOGN, LOGE, LAZY>), dim3(nb * L * 2), dim3(G::T), G::LDS_USED, st, a);
    if (ev) HX_CHECK(hipEventRecord(ev[3], st));
    return (int)hipGetLastError();
}

size_t hx_ks_f64_scratch_words(size_t L) { return L + (L + 1) * L + 2 * (L + 1) + 2; }   // per instance, in units of n

int hx_launch_keyswitch_f64(hexl_ks_plan* p, u64* d_result, const u64* d_t_target, size_t nb, int stage_mask,
                            hipEvent_t* ev) {
    const size_t n = p->n, L = p->L;
    KsArgsF a;
    a.mods = p->d_mods_f64; a.tables = p->d_tables_f64; a.keys = p->d_keys_f64;
    a.c = (double*)p->cur_scratch;
    a.u = a.c + p->cap * L * n;
    a.prod = a.u + p->cap * (L + 1) * L * n;
    a.s = a.prod + p->cap * 2 * (L + 1) * n;
    a.t_target = d_t_target; a.result = d_result;
    a.L = (u32)L; a.K = p->K; a.nb = (u32)nb;
    a.range_flag = p->d_flag;
    a.overwrite = p->overwrite_result ? 1u : 0u;
    a.skip = p->x_skip ? 1u : 0u;
    a.tiermap = 0;
    for (u32 i = 0; i < p->K; ++i) a.tiermap |= (unsigned long long)(p->tier[i] & 15u) << (4 * i);
    // Limbs of different tiers: the kernels built with LAZY = -1 look the schedule up per transform (with_tier). They are NOT the default
    // here: this pipeline serves the batches that do not fill the chip, where latency binds, not FP64 issue, and a kernel that carries
    // two to four copies of its transforms spills (k_ksf_moddown 80 registers, k_ksl_up 116) -- bridge-seal's chain at 2 ... 48 instances runs
    // 1-12 % FASTER on the plan-wide tier (round 5, tools/seal_chain_rate.py; 4 instances: 44.3 k against 38.9 k keyswitch/s). The slot-major
    // pipeline (one launch per tier group, +14 %) and the lone-keyswitch kernels keep their per-limb tiers. HEXL_KS_PER_LIMB=2 selects the
    // per-transform lookup here as well (tests).
    static const bool lookup = [] { const char* e = getenv("HEXL_KS_PER_LIMB"); return e && atoi(e) == 2; }();
    if (p->mixed && lookup) {
        switch (p->logn) {
            case 10: return run_chunk_f64<10, 4, -1>(p, a, stage_mask, ev);
            case 11: return run_chunk_f64<11, 5, -1>(p, a, stage_mask, ev);
            case 12: return run_chunk_f64<12, 5, -1>(p, a, stage_mask, ev);
            case 13: return run_chunk_f64<13, 5, -1>(p, a, stage_mask, ev);
            case 14: return run_chunk_f64<14, 4, -1>(p, a, stage_mask, ev);
            case 15: return run_chunk_f64<15, 5, -1>(p, a, stage_mask, ev);
            default: return HEXL_E_BADARG;
        }
    }
    // LAZY template argument = forward reduction period (f64_arith.hpp): 3 when every modulus <= 2^51(1+2^-7), 6 / 12
    // for moduli <= 2^50 / 2^49 (N = 16384 only; the smaller transforms keep 3), 0 = strict
    if (p->f64_lazy) {
        switch (p->logn) {
            case 10: return run_chunk_f64<10, 4, 3>(p, a, stage_mask, ev);
            case 11: return run_chunk_f64<11, 5, 3>(p, a, stage_mask, ev);
            case 12: return run_chunk_f64<12, 5, 3>(p, a, stage_mask, ev);
            case 13: return run_chunk_f64<13, 5, 3>(p, a, stage_mask, ev);
            case 15: return run_chunk_f64<15, 5, 3>(p, a, stage_mask, ev);      // beyond the reference: N = 32768
            case 14: return p->f64_lazy == 12 ? run_chunk_f64<14, 4, 12>(p, a, stage_mask, ev)
                          : p->f64_lazy == 6 ? run_chunk_f64<14, 4, 6>(p, a, stage_mask, ev)
                                             : run_chunk_f64<14, 4, 3>(p, a, stage_mask, ev);
            default: return HEXL_E_BADARG;
        }
    }
    switch (p->logn) {
        case 10: return run_chunk_f64<10, 4, 0>(p, a, stage_mask, ev);
        case 11: return run_chunk_f64<11, 5, 0>(p, a, stage_mask, ev);
        case 12: return run_chunk_f64<12, 5, 0>(p, a, stage_mask, ev);
        case 13: return run_chunk_f64<13, 5, 0>(p, a, stage_mask, ev);
        case 15: return run_chunk_f64<15, 5, 0>(p, a, stage_mask, ev);
        case 14: return run_chunk_f64<14, 4, 0>(p, a, stage_mask, ev);
        default: return HEXL_E_BADARG;
    }
}
