// keyswitch.hip -- K4: the CKKS key-switch pipeline for gfx950 as three fused kernels.
// Replaces the autorun dataflow of device/keyswitch/ (load.hpp -> intt1 -> intt1_redu -> ntt1 ->
// dyadmult -> intt2 -> intt2_redu -> ntt2 -> ms -> store.hpp; wiring in
// autorun_kernel_instances.hpp:43-255) plus the host-side accumulate of fpga.cpp:441-475.
//
//   ks_intt    (b, d)          c_d = INTT_{q_d}(t_target[d])                      [SURVEY K4 step 1]
//   ks_modup   (b, slot)       for d: u = NTT_{q_i}(c_d mod q_i); acc_k += u . key[d][k][i]
//                              slot < L : prod[k][i] = acc_k                      [steps 2-3]
//                              slot = L (special prime): s'_k = INTT(acc_k) + floor(q_sp/2)   [step 4]
//   ks_moddown (b, i, k)       w = NTT_{q_i}((s'_k + fix_i) mod q_i);
//                              result[k][i] += (prod[k][i] - w) * msf_i  (mod q_i) [steps 5-7]
//
// All arithmetic is canonical in the reference (AddUIntMod / SubUIntMod / MultiplyUIntMod), so any
// exact evaluation matches bit for bit; here the transforms reuse the Harvey lazy butterflies of
// ntt_core.hpp with a final reduction to [0,q). Intermediates that never leave the GPU (prod, keys)
// are kept in the forward transform's register order ("B order", [r][tid]) so they are read and
// written with fully coalesced accesses.
#include <stdlib.h>

// wave priority by pass of the integer forward transforms (ntt_core.hpp WgNtt::prio; keyswitch_x.hip has the reasoning):
// 92.2 k -> 93.8 k keyswitch/s on the integer kernels at L = 7
#ifndef HX_IFWD_PRIO
#define HX_IFWD_PRIO 1222
#endif
#include "hexl_internal.hpp"
#include "ntt_core.hpp"

using namespace hx;

// ---- small modular helpers (device/mod_ops.hpp:206-224) -------------------------------------
__device__ __forceinline__ u64 barrett64(u64 v, u64 q, u64 qbarr) {      // BarrettReduce64 :213-217
    return csub(v - mulhi(v, qbarr) * q, q);
}
__device__ __forceinline__ u64 mulmod128(u64 x, u64 y, u64 q, u32 len, u64 barr_lo) {
    const u64 lo = x * y, hi = mulhi(x, y);
    const u64 c1 = (lo >> len) | (hi << (64 - len));                     // q >= 2^16 so len >= 15
    return csub(lo - mulhi(c1, barr_lo) * q, q);
}

// Observed dispatch puts block b on XCD b % 8 (MI355X_MICROARCH: workgroup dispatch). Give every XCD a
// contiguous range of work items so blocks sharing one L2 work on the same RNS slot (same key slabs
// and twiddle tables). Bijective for any total; affects speed only.
__device__ __forceinline__ u32 xcd_item(u32 bid, u32 total) {
    const u32 q = total >> 3, r = total & 7, xcd = bid & 7, j = bid >> 3;
    return xcd * q + (xcd < r ? xcd : r) + j;
}

struct KsArgs {
    const KsModulus* mods;   // [K]
    const u64* tables;       // [K][4][n]
    const u64* keys;         // [L][L+1][2][2][n]  B order; [..][k][0] = key mod q, [..][k][1] = its Shoup factor floor(key 2^64 / q)
    u64* c;                  // [chunk][L][n]      natural order
    u64* prod;               // [chunk][2][L][n]   B order
    u64* s;                  // [chunk][2][n]      natural order
    const u64* t_target;     // [chunk][L][n]
    u64* result;             // [chunk][2][L][n]
    u32 L, K, nb;            // nb = instances in this chunk
};

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ks_intt(KsArgs a) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const u32 item = blockIdx.x;                      // item = b*L + d
    const u32 d = __builtin_amdgcn_readfirstlane(item % a.L);
    const KsModulus md = a.mods[d];
    const u64* tb = a.tables + size_t(d) * 4 * G::N;
    const u64* src = a.t_target + size_t(item) * G::N;
    u64 v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = src[G::idxB(r, tid)];
    WgNtt<LOGN, LOGE>::template inverse<true>(v, lds, tid, tb + 2 * G::N, tb + 3 * G::N, md.q, md.inv_n, md.inv_n_p,
                                              md.inv_n_w, md.inv_n_w_p);
    u64* dst = a.c + size_t(item) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r) dst[G::idxA(r, tid)] = v[r];
}

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ks_modup(KsArgs a) {
    using G = Geom<LOGN, LOGE>;
    using W = WgNtt<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const u32 L = a.L;
    const u32 item = xcd_item(blockIdx.x, gridDim.x);  // slot-major: item = slot*nb + b
    // integer division runs on the VALU; pin the (uniform) results back into SGPRs so table bases stay scalar
    const u32 slot = __builtin_amdgcn_readfirstlane(item / a.nb), b = item - slot * a.nb;
    const u32 i = slot < L ? slot : a.K - 1;            // special prime = moduli[K-1] (load.hpp:79-84)
    const KsModulus md = a.mods[i];
    const u64 q = md.q;
    const u32 len = u32(md.len);
    const u64* tb = a.tables + size_t(i) * 4 * G::N;

    u64 acc0[G::E], acc1[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) acc0[r] = acc1[r] = 0;

    for (u32 d = 0; d < L; ++d) {
        const u64* cd = a.c + (size_t(b) * L + d) * G::N;
        u64 v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = barrett64(cd[G::idxA(r, tid)], q, md.qbarr);   // intt1_redu.hpp:36-42
        const u64* roots = tb + opaque_zero();          // keep twiddle loads inside the d loop
        W::forward_lazy(v, lds, tid, roots, roots + G::N, q);
        W::final_reduce(v, q);
        const u64* k0 = a.keys + ((size_t(d) * (L + 1) + slot) * 4) * G::N;
        const u64* k1 = k0 + 2 * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) {                 // dyadmult.hpp:128-140
            const u64 key0 = k0[r * G::T + tid], key1 = k1[r * G::T + tid];
            acc0[r] = csub(acc0[r] + mulmod128(v[r], key0, q, len, md.barr_lo), q);
            acc1[r] = csub(acc1[r] + mulmod128(v[r], key1, q, len, md.barr_lo), q);
        }
    }

    if (slot < L) {
        u64* p0 = a.prod + ((size_t(b) * 2 + 0) * L + slot) * G::N;
        u64* p1 = a.prod + ((size_t(b) * 2 + 1) * L + slot) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) { p0[r * G::T + tid] = acc0[r]; p1[r * G::T + tid] = acc1[r]; }
    } else {
        // special-prime limb: INTT, then + floor(q_sp/2) mod q_sp (intt2_redu.hpp:25,43)
        const u64* it = tb + opaque_zero() + 2 * G::N;
        W::template inverse<false, OrderedByCaller>(acc0, lds, tid, it, it + G::N, q, md.inv_n, md.inv_n_p, md.inv_n_w, md.inv_n_w_p);   // (behind forward transforms)
        u64* s0 = a.s + (size_t(b) * 2 + 0) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) s0[G::idxA(r, tid)] = csub(acc0[r] + md.half, q);
        it = tb + opaque_zero() + 2 * G::N;
        __syncthreads();                                // inverse after inverse: the first one's cross-wave readers (ntt_core.hpp ReadersGate; once per instance here)
        W::template inverse<false, OrderedByCaller>(acc1, lds, tid, it, it + G::N, q, md.inv_n, md.inv_n_p, md.inv_n_w, md.inv_n_w_p);   // (behind the barrier above)
        u64* s1 = a.s + (size_t(b) * 2 + 1) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) s1[G::idxA(r, tid)] = csub(acc1[r] + md.half, q);
    }
}

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ks_moddown(KsArgs a) {
    using G = Geom<LOGN, LOGE>;
    using W = WgNtt<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const u32 L = a.L;
    const u32 item = xcd_item(blockIdx.x, gridDim.x);  // limb-major: item = (i*2 + k)*nb + b
    const u32 ik = __builtin_amdgcn_readfirstlane(item / a.nb), b = item - ik * a.nb;
    const u32 i = ik >> 1, k = ik & 1;
    const KsModulus md = a.mods[i];
    const u64 q = md.q;
    const u64* tb = a.tables + size_t(i) * 4 * G::N;

    const u64* sk = a.s + (size_t(b) * 2 + k) * G::N;
    u64 v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = barrett64(sk[G::idxA(r, tid)] + md.fix, q, md.qbarr);   // intt2_redu.hpp:49-51
    W::template forward_lazy<true>(v, lds, tid, tb, tb + G::N, q);
    W::final_reduce(v, q);

    const u64* pk = a.prod + ((size_t(b) * 2 + k) * L + i) * G::N;
    u64* res = a.result + ((size_t(b) * 2 + k) * L + i) * G::N;
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const u64 in = csub(pk[r * G::T + tid] + q - v[r], q);                   // ms.hpp:70-78 (canonical)
        const u64 out = csub(lazy_mul(in, md.msf, md.msf_p, q), q);              // ms.hpp:80-82
        const int idx = G::idxB(r, tid);
        const u64 rr = res[idx] + out;                                           // fpga.cpp:453-457
        res[idx] = rr >= q ? rr - q : rr;
    }
}

// ---- second generation (HEXL_KS_PIPE != 1): prod[k][i] never leaves the registers -------------------------------------
// The layout of keyswitch_x.hip with the integer butterflies: one workgroup per (instance, limb) carries the two
// accumulators through steps 2-3 AND 5-7 (`prod` and the k_ks_moddown launch are gone), the special slot has its own
// kernel, the d == i round uses t_target[i] itself (NTT(INTT(t_i)) = t_i for in-range data), and every
// multiply-accumulate requests the next round's input into the registers its products free.
//
//   k_ks_intt     (b, d)    c_d = INTT_{q_d}(t_target[d])                                           (step 1, as above)
//   k_ksi_special (b)       acc_k = sum_d NTT_{q_sp}(c_d mod q_sp) . key[d][special][k];  s'_k = INTT(acc_k) + floor(q_sp/2)
//   k_ksi_main    (b, i<L)  acc_k = sum_d NTT_{q_i}(c_d mod q_i) . key[d][i][k]; then for k = 0, 1:
//                           w = NTT_{q_i}((s'_k + fix_i) mod q_i);  result[k][i] += (acc_k - w) . msf_i

// acc_k += v . key_k (dyadmult.hpp:128-140); v[r] is replaced by word r of `next` (A order, never null).
// The keys carry their Shoup factors (computed once per key set on the host), so a term is one Harvey lazy product --
// ~20 VALU instructions against ~50 for the 128 -> 64-bit Barrett product of the first generation -- and takes the
// transform's output as it is (v < 4q; lazy_mul_n accepts any 64-bit x and returns < 2q). The accumulators stay in [0, 2q).
template <class G>
__device__ __forceinline__ void mac_keys_i(u64 (&acc0)[G::E], u64 (&acc1)[G::E], u64 (&v)[G::E], const u64* __restrict__ k0,
                                           const u64* __restrict__ next, int tid, const KsModulus& md) {
    constexpr int PF = G::E >= 32 ? 4 : 2;                  // key ring depth (deeper rings cost registers: 80.7 k against 92 k keyswitch/s at eight)
    const RowStream<u64> keys(k0, 4 * G::N * 8), nxt(next, G::N * 8);
    const u32 toff = u32(tid) * 8;
    const ModConst mc = mod_const(md.q);
    u64 ka[PF], kap[PF], kb[PF], kbp[PF];
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        ka[r] = keys.at(toff, r * G::T * 8);                 kap[r] = keys.at(toff, (G::N + r * G::T) * 8);
        kb[r] = keys.at(toff, (2 * G::N + r * G::T) * 8);    kbp[r] = keys.at(toff, (3 * G::N + r * G::T) * 8);
    }
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const u64 ca = ka[r % PF], cap = kap[r % PF], cb = kb[r % PF], cbp = kbp[r % PF];
        if (r + PF < G::E) {
            ka[r % PF] = keys.at(toff, (r + PF) * G::T * 8);                kap[r % PF] = keys.at(toff, (G::N + (r + PF) * G::T) * 8);
            kb[r % PF] = keys.at(toff, (2 * G::N + (r + PF) * G::T) * 8);   kbp[r % PF] = keys.at(toff, (3 * G::N + (r + PF) * G::T) * 8);
        }
        const u64 x = v[r];
        v[r] = nxt.at(toff, G::idxA(r, 0) * 8);
        acc0[r] = csub_n(acc0[r] + lazy_mul_n(x, ca, cap, mc.nq), mc.twoq, mc.n2q);
        acc1[r] = csub_n(acc1[r] + lazy_mul_n(x, cb, cbp, mc.nq), mc.twoq, mc.n2q);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// accumulators of mac_keys_i back to [0, q)
template <int E>
__device__ __forceinline__ void canonical(u64 (&acc)[E], u64 q) {
    const ModConst mc = mod_const(q);
#pragma unroll
    for (int r = 0; r < E; ++r) acc[r] = csub_n(acc[r], q, mc.nq);
}

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksi_special(KsArgs a) {
    using G = Geom<LOGN, LOGE>;
    using W = WgNtt<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 L = a.L, b = blockIdx.x, isp = a.K - 1;
    const KsModulus md = a.mods[isp];
    const u64 q = md.q;
    u64 acc0[G::E], acc1[G::E], v[G::E];                          // v between rounds: the next round's input, A order
    {
        const int tid = threadIdx.x;
        const u64* c0 = a.c + size_t(b) * L * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) { acc0[r] = 0; acc1[r] = 0; v[r] = (c0 + G::idxA(r, 0))[u32(tid)]; }
    }
#pragma unroll 1
    for (u32 d = 0; d < L; ++d) {
        int tid = threadIdx.x;                                    // laundered per round: nothing is hoisted out of the loop
        asm volatile("" : "+v"(tid));
        const u64* ts = a.tables + size_t(isp) * 4 * G::N + opaque_zero();
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = barrett64(v[r], q, md.qbarr);             // intt1_redu.hpp:36-42
        W::forward_lazy(v, lds, tid, ts, ts + G::N, q);            // v < 4q
        const u32 nd = d + 1 < L ? d + 1 : d;                     // (the last limb is requested twice: harmless)
        mac_keys_i<G>(acc0, acc1, v, a.keys + ((size_t(d) * (L + 1) + L) * 4) * G::N, a.c + (size_t(b) * L + nd) * G::N, tid, md);
    }
    canonical(acc0, q);
    canonical(acc1, q);
    // s'_k = INTT(acc_k) + floor(q_sp/2) (mod q_sp)   (intt2_redu.hpp:25,43)
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u64* it = a.tables + size_t(isp) * 4 * G::N + opaque_zero() + 2 * G::N;
        W::template inverse<false, OrderedByCaller>(acc0, lds, tid, it, it + G::N, q, md.inv_n, md.inv_n_p, md.inv_n_w, md.inv_n_w_p);   // (behind forward transforms)
        u64* s0 = a.s + (size_t(b) * 2 + 0) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) (s0 + G::idxA(r, 0))[u32(tid)] = csub(acc0[r] + md.half, q);
    }
    __syncthreads();                                    // inverse after inverse: the first one's cross-wave readers (ntt_core.hpp ReadersGate; once per instance here)
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u64* it = a.tables + size_t(isp) * 4 * G::N + opaque_zero() + 2 * G::N;
        W::template inverse<false, OrderedByCaller>(acc1, lds, tid, it, it + G::N, q, md.inv_n, md.inv_n_p, md.inv_n_w, md.inv_n_w_p);   // (behind the barrier above)
        u64* s1 = a.s + (size_t(b) * 2 + 1) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) (s1 + G::idxA(r, 0))[u32(tid)] = csub(acc1[r] + md.half, q);
    }
}

// steps 5-7 for one k: v holds the raw s'_k words (A order); result[k][i] += (acc - NTT((s'_k + fix_i) mod q_i)) . msf_i
template <class G, class W>
__device__ __forceinline__ void ksi_down_round(u64 (&v)[G::E], const u64 (&acc)[G::E], u64* __restrict__ res, u64* lds, int tid,
                                               const u64* tb, const KsModulus& md) {
    const u64 q = md.q;
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = barrett64(v[r] + md.fix, q, md.qbarr);       // intt2_redu.hpp:49-51
    W::forward_lazy(v, lds, tid, tb, tb + G::N, q);
    W::final_reduce(v, q);
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const u64 in = csub(acc[r] + q - v[r], q);                                      // ms.hpp:70-78 (canonical)
        v[r] = csub(lazy_mul(in, md.msf, md.msf_p, q), q);                              // ms.hpp:80-82
    }
    // read-modify-write at the thread's own (B) positions, half of the registers at a time (fpga.cpp:453-457)
    const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
    for (int r0 = 0; r0 < G::E; r0 += G::E / 2) {
        u64 old[G::E / 2];
#pragma unroll
        for (int r = 0; r < G::E / 2; ++r) old[r] = (res + G::idxB(r0 + r, 0))[tB];
#pragma unroll
        for (int r = 0; r < G::E / 2; ++r) {
            const u64 rr = old[r] + v[r0 + r];
            (res + G::idxB(r0 + r, 0))[tB] = rr >= q ? rr - q : rr;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ksi_main(KsArgs a) {
    using G = Geom<LOGN, LOGE>;
    using W = WgNtt<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 L = a.L;
    // instance-major, XCD-contiguous: the L workgroups that read the same c_d and s' run side by side on one XCD
    const u32 item = __builtin_amdgcn_readfirstlane(xcd_item(blockIdx.x, gridDim.x));
    const u32 b = item / L, i = item - b * L;
    const KsModulus md = a.mods[i];
    const u64 q = md.q;
    // round `it` reads c_it (it < L, skipping it == i) or s'_{it-L}
    auto round_src = [&](u32 it) { return it < L ? a.c + (size_t(b) * L + it) * G::N : a.s + (size_t(b) * 2 + (it - L)) * G::N; };
    const u32 first = i == 0 ? 1u : 0u;
    u64 acc0[G::E], acc1[G::E], v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) { acc0[r] = 0; acc1[r] = 0; }
    {
        // d == i: NTT_{q_i}(INTT_{q_i}(t_i) mod q_i) = t_i mod q_i (the reference recomputes it)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u64* src = a.t_target + (size_t(b) * L + i) * G::N;
        const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = barrett64((src + G::idxB(r, 0))[tB], q, md.qbarr);
        mac_keys_i<G>(acc0, acc1, v, a.keys + ((size_t(i) * (L + 1) + i) * 4) * G::N, round_src(first), tid, md);
    }
#pragma unroll 1
    for (u32 it = first; it < L;) {                                // rounds d != i
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u64* tb = a.tables + size_t(i) * 4 * G::N + opaque_zero();
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = barrett64(v[r], q, md.qbarr);             // intt1_redu.hpp:36-42
        u32 nit = it + 1;
        if (nit == i) ++nit;
        W::forward_lazy(v, lds, tid, tb, tb + G::N, q);            // v < 4q
        mac_keys_i<G>(acc0, acc1, v, a.keys + ((size_t(it) * (L + 1) + i) * 4) * G::N, round_src(nit), tid, md);   // nit <= L
        it = nit;
    }
    canonical(acc0, q);
    canonical(acc1, q);
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u64* tb = a.tables + size_t(i) * 4 * G::N + opaque_zero();
        ksi_down_round<G, W>(v, acc0, a.result + ((size_t(b) * 2 + 0) * L + i) * G::N, lds, tid, tb, md);
        const u64* nxt = a.s + (size_t(b) * 2 + 1) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = (nxt + G::idxA(r, 0))[u32(tid)];
    }
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u64* tb = a.tables + size_t(i) * 4 * G::N + opaque_zero();
        ksi_down_round<G, W>(v, acc1, a.result + ((size_t(b) * 2 + 1) * L + i) * G::N, lds, tid, tb, md);
    }
}

template <int LOGN, int LOGE>
static int run_chunk_i(hexl_ks_plan* p, const KsArgs& a, int stage_mask, hipEvent_t* ev) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc = once.run(p->ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ks_intt<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_USED));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ksi_special<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_USED));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ksi_main<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_USED));
            return 0;
        }))
        return rc;
    hipStream_t st = p->cur;
    if (ev) HX_CHECK(hipEventRecord(ev[0], st));
    if (stage_mask & 1)
        hipLaunchKernelGGL((k_ks_intt<LOGN, LOGE>), dim3(a.nb * a.L), dim3(G::T), G::LDS_USED, st, a);
    if (ev) HX_CHECK(hipEventRecord(ev[1], st));
    if (stage_mask & 2)
        hipLaunchKernelGGL((k_ksi_special<LOGN, LOGE>), dim3(a.nb), dim3(G::T), G::LDS_USED, st, a);
    if (ev) HX_CHECK(hipEventRecord(ev[2], st));
    if (stage_mask & 4)
        hipLaunchKernelGGL((k_ksi_main<LOGN, LOGE>), dim3(a.nb * a.L), dim3(G::T), G::LDS_USED, st, a);
    if (ev) HX_CHECK(hipEventRecord(ev[3], st));
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
template <int LOGN, int LOGE>
static int run_chunk(hexl_ks_plan* p, const KsArgs& a, int stage_mask, hipEvent_t* ev) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc = once.run(p->ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ks_intt<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_USED));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ks_modup<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)G::LDS_USED));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ks_moddown<LOGN, LOGE>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED));
            return 0;
        }))
        return rc;
    hipStream_t st = p->cur;
    if (ev) HX_CHECK(hipEventRecord(ev[0], st));
    if (stage_mask & 1)
        hipLaunchKernelGGL((k_ks_intt<LOGN, LOGE>), dim3(a.nb * a.L), dim3(G::T), G::LDS_USED, st, a);
    if (ev) HX_CHECK(hipEventRecord(ev[1], st));
    if (stage_mask & 2)
        hipLaunchKernelGGL((k_ks_modup<LOGN, LOGE>), dim3(a.nb * (a.L + 1)), dim3(G::T), G::LDS_USED, st, a);
    if (ev) HX_CHECK(hipEventRecord(ev[2], st));
    if (stage_mask & 4)
        hipLaunchKernelGGL((k_ks_moddown<LOGN, LOGE>), dim3(a.nb * a.L * 2), dim3(G::T), G::LDS_USED, st, a);
    if (ev) HX_CHECK(hipEventRecord(ev[3], st));
    return (int)hipGetLastError();
}

// instances per scratch chunk: HEXL_KS_CHUNK, else 256 at N = 16384 and the same number of COEFFICIENTS per chunk at the
// other ring dimensions (4096 instances at N = 1024: a chunk's kernels must fill the chip whatever the transform size)
static size_t ks_chunk_default(const hexl_ks_plan* p) {
    static const long v = [] {
        const char* e = getenv("HEXL_KS_CHUNK");
        const long c = e ? atol(e) : 0;
        return c < 0 ? 0L : c;
    }();
    if (v) return (size_t)v;
    return p->logn >= 14 ? size_t(256) >> (p->logn - 14) : size_t(256) << (14 - p->logn);
}

size_t hx_ks_chunk(const hexl_ks_plan* p) { return ks_chunk_default(p); }
size_t hx_ks_f64_scratch_words(size_t L);
static size_t scratch_words(const hexl_ks_plan* p) {           // per instance, in units of n 64-bit words
    return p->use_f64 ? hx_ks_f64_scratch_words(p->L) : size_t(3) * p->L + 2;
}
int hx_ks_lanes() {
    static const int v = [] { const char* e = getenv("HEXL_KS_LANES"); const int l = e ? atoi(e) : 2; return l < 2 ? 2 : l > HX_KS_MAX_LANES ? HX_KS_MAX_LANES : l; }();
    return v;
}
size_t hexl_ks_scratch_bytes(const hexl_ks_plan* p, size_t batch) {
    const size_t chunk = batch < ks_chunk_default(p) ? batch : ks_chunk_default(p);
    return size_t(hx_ks_lanes()) * chunk * scratch_words(p) * p->n * sizeof(u64);      // two lanes
}

// HEXL_KS_VALIDATE=1: the keyswitch precondition (every t_target / result word below its modulus), checked on the device
__global__ void k_ks_validate(const u64* __restrict__ t, const u64* __restrict__ res, const KsModulus* __restrict__ mods,
                              u32 L, u32 n, size_t batch, u32* __restrict__ bad) {
    // res == nullptr: the call WRITES `result` (host-pointer path on the overwrite kernels): only t_target is an input
    const size_t per_t = size_t(L) * n, per_r = 2 * per_t, total = batch * (per_t + (res ? per_r : 0));
    bool out = false;
    for (size_t g = size_t(blockIdx.x) * blockDim.x + threadIdx.x; g < total; g += size_t(gridDim.x) * blockDim.x) {
        u64 word; u32 limb;
        if (g < batch * per_t) { word = t[g]; limb = u32((g % per_t) / n); }
        else { const size_t h = g - batch * per_t; word = res[h]; limb = u32(((h % per_r) / n) % L); }
        out |= word >= mods[limb].q;
    }
    if (out) atomicOr(bad, 1u);
}

static int validate_inputs(hexl_ks_plan* p, const u64* d_result, const u64* d_t_target, size_t batch) {
    // HEXL_KS_VALIDATE has a flag word of its own beside the kernels' range flag (d_flag[1] / h_flag[1], allocated with the plan:
    // nothing to allocate, free or leak per call, no asynchronous copy into pageable memory): HEXL_E_RANGE means "THIS call's inputs
    // are out of range, nothing was computed". A violation an earlier launch on the plan has flagged in d_flag[0] and nobody has
    // read yet stays there for hexl_ks_range_check / hexl_keyswitch_host (HEXL_W_RANGE) -- it is neither reported as a refusal of
    // this, valid, call nor wiped (the shared word conflated the two statuses).
    // A call that WRITES its result (overwrite_result: the staging buffer of the host-pointer path is uninitialised on
    // purpose) has only t_target as input.
    HX_CHECK(hipMemsetAsync(p->d_flag + 1, 0, sizeof(u32), p->ctx->stream));
    hipLaunchKernelGGL(k_ks_validate, dim3(2048), dim3(256), 0, p->ctx->stream, d_t_target, p->overwrite_result ? nullptr : d_result,
                       p->d_mods, p->L, p->n, batch, p->d_flag + 1);
    HX_CHECK(hipMemcpyAsync(p->h_flag + 1, p->d_flag + 1, sizeof(u32), hipMemcpyDeviceToHost, p->ctx->stream));
    HX_CHECK(hipStreamSynchronize(p->ctx->stream));
    return p->h_flag[1] ? HEXL_E_RANGE : 0;
}

// the (b, d)-major FP64 kernels can write `result` instead of accumulating into it: every chunk of the batch must take them
bool hx_ks_can_overwrite(const hexl_ks_plan* p, size_t nb) {
    if (!p->use_f64 || p->logn > 14) return false;               // (N = 32768: no registers for a second epilogue)
    const size_t chunk = nb < ks_chunk_default(p) ? nb : ks_chunk_default(p);
    return !hx_ks_x_applies(p, chunk) && !hx_ks_x_applies(p, nb % chunk ? nb % chunk : chunk);
}

int hx_launch_keyswitch(hexl_ks_plan* p, u64* d_result, const u64* d_t_target, size_t batch, int stage_mask,
                        hipEvent_t* ev) {
    if (!batch) return 0;
    if (!p->have_keys) return HEXL_E_NOKEYS;
    static const bool validate = [] { const char* e = getenv("HEXL_KS_VALIDATE"); return e && atoi(e) == 1; }();
    if (validate)
        if (int rc = validate_inputs(p, d_result, d_t_target, batch)) return rc;
    // chunks alternate between two lanes; a batch that fits one chunk is still split in two when it is large
    // enough to fill the chip twice, so the lanes always have something to overlap. Timing runs (ev) stay on one lane.
    size_t chunk = batch < ks_chunk_default(p) ? batch : ks_chunk_default(p);
    // (FP64 path: with one barrier per transform and steps 1-2 fused, two lanes measure the same as one stream
    // (160 k vs 162 k keyswitch/s), so it runs its chunks back to back on the caller's stream; HEXL_KS_ONE_LANE=0
    // brings the lanes back)
    // The slot-major pipeline (keyswitch_x.hip) alternates its chunks between the two lanes again: one chunk's three
    // kernels fill the ragged end of the other's last generation of workgroups (174 k against 160-170 k keyswitch/s).
    const char* lane_env = getenv("HEXL_KS_ONE_LANE");
    const bool one_lane = lane_env ? atoi(lane_env) == 1 : (p->use_f64 && !hx_ks_x_applies(p, chunk));
    const bool two_lanes = !ev && batch >= 64 && !one_lane && !(p->use_f64 && batch <= chunk);
    if (two_lanes && batch <= chunk) chunk = (batch + 1) / 2;
    const size_t lane_words = chunk * scratch_words(p) * p->n;
    if (p->cap < chunk) {
        if (p->d_scratch) { HX_CHECK(hipDeviceSynchronize()); HX_CHECK(hipFree(p->d_scratch)); }
        p->d_scratch = nullptr; p->cap = 0;
        HX_CHECK(hipMalloc((void**)&p->d_scratch, size_t(hx_ks_lanes()) * lane_words * sizeof(u64)));
        p->cap = chunk;
    }
    if (!p->aux[0]) {
        for (int l = 0; l < hx_ks_lanes(); ++l) {
            HX_CHECK(hipStreamCreateWithFlags(&p->aux[l], hipStreamNonBlocking));
            HX_CHECK(hipEventCreateWithFlags(&p->ev_done[l], hipEventDisableTiming));
        }
        HX_CHECK(hipEventCreateWithFlags(&p->ev_start, hipEventDisableTiming));
    }
    hipStream_t user = p->ctx->stream;
    const int lanes = two_lanes ? hx_ks_lanes() : 1;
    if (lanes >= 2) {                                                 // lanes start after everything queued so far
        HX_CHECK(hipEventRecord(p->ev_start, user));
        for (int l = 0; l < lanes; ++l) HX_CHECK(hipStreamWaitEvent(p->aux[l], p->ev_start, 0));
    }
    const size_t n = p->n, L = p->L;
    // lane 1's first chunk is half-sized (lane l's: l / lanes): the lanes then run out of phase, so one lane's HBM-bound kernels overlap
    // the other's FP64-bound ones instead of both running the same kernel side by side
    size_t ci = 0, nb = 0;
    for (size_t b0 = 0; b0 < batch; b0 += nb, ++ci) {
        const size_t want = (lanes >= 2 && ci >= 1 && ci < (size_t)lanes && !p->use_f64) ? (chunk * ci + lanes - 1) / lanes : chunk;
        nb = (batch - b0 < want) ? batch - b0 : want;
        const int lane = (int)(ci % lanes);
        p->cur = lanes >= 2 ? p->aux[lane] : user;                   // one lane: straight on the caller's stream
        p->cur_scratch = p->d_scratch + size_t(lane) * p->cap * scratch_words(p) * n;
        int rc;
        if (p->use_f64 && hx_ks_x_applies(p, nb)) {
            rc = hx_launch_keyswitch_x(p, d_result + b0 * 2 * L * n, d_t_target + b0 * L * n, nb, stage_mask, ev);
            if (rc) return rc;
            continue;
        }
        if (p->use_f64 && !ev && stage_mask == 7 && hx_ks_lat_applies(p, nb)) {     // a lone keyswitch: every transform cut in four
            if ((rc = hx_launch_keyswitch_lat(p, d_result + b0 * 2 * L * n, d_t_target + b0 * L * n, nb))) return rc;
            continue;
        }
        if (p->use_f64) {
            rc = hx_launch_keyswitch_f64(p, d_result + b0 * 2 * L * n, d_t_target + b0 * L * n, nb, stage_mask, ev);
            if (rc) return rc;
            continue;
        }
        KsArgs a;
        a.mods = p->d_mods; a.tables = p->d_tables; a.keys = p->d_keys;
        a.c = p->cur_scratch;
        a.prod = a.c + p->cap * L * n;
        a.s = a.prod + p->cap * 2 * L * n;
        a.t_target = d_t_target + b0 * L * n;
        a.result = d_result + b0 * 2 * L * n;
        a.L = (u32)L; a.K = p->K; a.nb = (u32)nb;
        // HEXL_KS_PIPE=1: the first-generation kernels (k_ks_modup / k_ks_moddown, `prod` through memory)
        static const bool gen1 = [] { const char* e = getenv("HEXL_KS_PIPE"); return e && atoi(e) == 1; }();
        switch (p->logn * 8 + p->int_loge) {
            case 10 * 8 + 4: rc = gen1 ? run_chunk<10, 4>(p, a, stage_mask, ev) : run_chunk_i<10, 4>(p, a, stage_mask, ev); break;
            case 11 * 8 + 5: rc = gen1 ? run_chunk<11, 5>(p, a, stage_mask, ev) : run_chunk_i<11, 5>(p, a, stage_mask, ev); break;
            case 12 * 8 + 5: rc = gen1 ? run_chunk<12, 5>(p, a, stage_mask, ev) : run_chunk_i<12, 5>(p, a, stage_mask, ev); break;
            case 13 * 8 + 5: rc = gen1 ? run_chunk<13, 5>(p, a, stage_mask, ev) : run_chunk_i<13, 5>(p, a, stage_mask, ev); break;
            case 14 * 8 + 5: rc = gen1 ? run_chunk<14, 5>(p, a, stage_mask, ev) : run_chunk_i<14, 5>(p, a, stage_mask, ev); break;
            case 14 * 8 + 4: rc = gen1 ? run_chunk<14, 4>(p, a, stage_mask, ev) : run_chunk_i<14, 4>(p, a, stage_mask, ev); break;
            default: rc = HEXL_E_BADARG;
        }
        if (rc) return rc;
    }
    for (int l = 0; lanes >= 2 && l < lanes; ++l) {                    // the caller's stream continues after both lanes
        HX_CHECK(hipEventRecord(p->ev_done[l], p->aux[l]));
        HX_CHECK(hipStreamWaitEvent(user, p->ev_done[l], 0));
    }
    return 0;
}

// fused ciphertext multiply + relinearize (the use-case of the reference's combined image,
// device/dyadic_multiply_keyswitch.cpp:4-5): chunks of the batch through the slot-major pipeline on the caller's stream
int hx_launch_mulrelin_x(hexl_ks_plan* p, u64* d_out, const u64* d_a, const u64* d_b, size_t nb);
int hx_launch_multiply_relinearize(hexl_ks_plan* p, u64* d_out, const u64* d_a, const u64* d_b, size_t batch) {
    if (!batch) return 0;
    if (!p->have_keys) return HEXL_E_NOKEYS;
    if (!p->use_f64 || p->logn < 10 || p->logn > 15) return HEXL_E_BADARG;   // see hx_launch_mulrelin_x
    const size_t chunk = batch < ks_chunk_default(p) ? batch : ks_chunk_default(p);
    const size_t lane_words = chunk * scratch_words(p) * p->n;
    if (p->cap < chunk) {
        if (p->d_scratch) { HX_CHECK(hipDeviceSynchronize()); HX_CHECK(hipFree(p->d_scratch)); }
        p->d_scratch = nullptr; p->cap = 0;
        HX_CHECK(hipMalloc((void**)&p->d_scratch, size_t(hx_ks_lanes()) * lane_words * sizeof(u64)));
        p->cap = chunk;
    }
    p->cur = p->ctx->stream;
    p->cur_scratch = p->d_scratch;
    const size_t per = 2 * p->L * p->n;
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        int rc = hx_launch_mulrelin_x(p, d_out + b0 * per, d_a + b0 * per, d_b + b0 * per, nb);
        if (rc) return rc;
    }
    return 0;
}
