// ntt.hip -- standalone batched forward / inverse negacyclic NTT kernels (K1, K2) for gfx950.
// Replaces device/fwd_ntt.cpp (fwd_ntt_kernel :82-497, ntt_input_kernel :499-580,
// ntt_output_kernel :582-604) and device/inv_ntt.cpp (:83-571) of the reference.
// One workgroup per polynomial; see ntt_core.hpp for the register/LDS mapping.
#include <stdlib.h>
#include <string.h>

// wave priority by pass of the FP64 forward transforms: the pass in front of the cross-wave barrier at 0, the rest at 1
// (keyswitch_x.hip has the reasoning; standalone forward NTT +5 % at batch 1024, +3.5 % at batch 4096; the inverse transforms'
// knob HX_INV_PRIO measured within +-2 % either way and stays off)
#ifndef HX_FWD_PRIO
#define HX_FWD_PRIO 1222
#endif
// ... and of the INTEGER inverse transforms (persistent k_ntt_inv_ip: moduli >= 2^52, tables that are not Shoup tables): the
// pass in front of the cross-wave barrier at 0, the others at 1: 8.9 M -> 9.7 M inverse NTT/s at q = 2^52 + 393217, batch 1024
// (1112: 9.1 M; the integer forward kernel runs one transform per workgroup and does not react to its knob)
#ifndef HX_IINV_PRIO
#define HX_IINV_PRIO 2212
#endif
#ifndef HX_IFWD_PRIO
#define HX_IFWD_PRIO 1222
#endif
#include "hexl_internal.hpp"
#include "ntt_core.hpp"
#include "ntt_core_f64.hpp"

// (N = 2048 -- 128-thread workgroups of two waves -- measured 3 % slower with the priorities: off there)
constexpr int ntt_fwd_prio(int logn) { return logn == 11 ? 0 : HX_FWD_PRIO; }

using namespace hx;

// k_ntt_inv_p: the per-lane twiddle pairs of an inverse pass requested before its butterflies (ntt_core_f64.hpp inv_stages_f64_pre, as in
// k_ksx_intt): +2.4 % inverse at batch 1024 (13.5 -> 13.85 M NTT/s, two rounds), +-0 at 4096. The forward counterpart (WgNttF64 PRE = 1 / 10 /
// 11, the keyswitch's early twiddle requests) measured within noise either way and stays off (profiles/r06_ab_ntt_pre.txt).
#ifndef NTT_IPRE
#define NTT_IPRE 1
#endif
// lazy tiers, N <= 16384: the forward transform on the X schedule of f64_arith.hpp (the added operand range-reduced where the bound chain needs
// it: 18 + the final 6 instead of 30 reduction instructions per butterfly column in the top tier). 0 = the periodic schedule.
#ifndef NTT_XSCHED
#define NTT_XSCHED 1
#endif
// ... and the inverse transform on the I schedule (range reductions by the history of a butterfly's inputs). 0 = every sum at every stage.
#ifndef NTT_ISCHED
#define NTT_ISCHED 1
#endif
// (the kernels' SEMI parameter: strict tier with the semi-strict butterflies in the WAVE-UNIFORM passes, ntt_core_f64.hpp SEMIU)

// ---------------------------------------------------------------------------------------------
// Exact-arithmetic fast path. The Harvey kernels above must be replayed op for op only where that is observable:
// out-of-range data, improper tables (benchmarks pass random ones), moduli >= 2^52. When q < 2^52, the tables
// satisfy precon[i] == floor(roots[i]*2^64/q) with roots[i] < q, and the polynomial is inside the algorithm's
// input range (< 4q forward, < 2q inverse), the reference's result is by construction the canonical transform
// of (x mod q) -- which the FP64 butterflies of ntt_core_f64.hpp compute exactly with ~1/3 of the instructions.
// k_ntt_prepare verifies the tables on the device (no division: 0 <= roots*2^64 - precon*q < q) and derives the
// centred double tables; the transform kernels then choose per polynomial, falling back to the integer
// butterflies whenever a precondition fails, so every input still gets the reference's exact answer.
// ---------------------------------------------------------------------------------------------
// one table entry i != 0: verify the Shoup pair (false: it is not one), derive the centred double(s)
__device__ __forceinline__ bool prepare_entry(const u64* __restrict__ roots, const u64* __restrict__ precon, u64 q, u32 i,
                                              double* w, double* wp) {
    const u64 r = roots[i], p = precon[i];
    // 128-bit  D = r*2^64 - p*q  must satisfy 0 <= D < q
    const u64 lo = p * q, hi = mulhi(p, q);
    const u64 d_lo = 0 - lo, d_hi = r - hi - (lo != 0);
    const bool ok = (r < q) && (hi + (lo != 0) <= r) && (d_hi == 0) && (d_lo < q);
    const double pd = (double)q;
    const u64 rr = r < q ? r : 0;
    const double c = rr > q / 2 ? (double)rr - pd : (double)rr;
    w[i] = c;          // (the lazy transforms take their quotients from the products: no w/p table, f64_arith.hpp)
    if (pd > hxf::LAZY_MAX_MODULUS) wp[i] = c / pd;                  // strict tier: the semi-strict forward schedule reads it
    return ok;
}

// What a launch needs to know about its tables. The counters come from rings of 64 slots (one per launch, round robin: no memset per
// launch -- the preparation of launch k zeroes the slots launch k + 32 will use, long after their last reader has finished on this stream).
struct NttPrep {
    double* w; double* wp;               // derived tables: written by k_ntt_prepare, read by the transforms (never restrict / const here)
    u32* viol; u32* viol_later;          // entries that are not Shoup pairs, this launch's slot / the slot zeroed for launch + 32
    u32 n;
    u32* redo;                           // launches with a k_ntt_redo_* behind them: [0] how many polynomials the integer butterflies must redo, [1 ...] which
};

__global__ void k_ntt_prepare(const u64* __restrict__ roots, const u64* __restrict__ precon, u64 q, NttPrep pr) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pr.n) return;
    // index 0 is never read by either transform; its thread resets the counters a launch 32 launches from now will use
    if (i == 0) {
        pr.w[0] = 0.0;
        *pr.viol_later = 0;
        if (pr.redo) *pr.redo = 0;
        return;
    }
    // (one atomic per wave that saw a bad entry, not one per entry: random tables -- the reference benchmark's -- fail in every entry)
    const bool ok = prepare_entry(roots, precon, q, i, pr.w, pr.wp);
    if (__builtin_amdgcn_ballot_w64(!ok) != 0 && (threadIdx.x & 63) == __builtin_amdgcn_readfirstlane(threadIdx.x & 63)) atomicAdd(pr.viol, 1u);
}

// The derived tables of a launch and whether k_ntt_prepare (a launch of its own in front of every fast-path launch: 2.8 us + a dispatch
// gap) found an entry that is not a Shoup pair. (The preparation INSIDE the persistent transform launch -- per-XCD table copies, ticketed
// slices -- was built in rounds 4-5, is bit-exact and measured no faster: tools/experiments/ntt_fused_prepare*.patch, README.md.)
__device__ __forceinline__ bool ntt_tables_ready(const NttPrep& pr, const double*& w, const double*& wp) {
    // (no laundering of the pointers: rounds 5's fused variant passed them through an asm barrier, which cost them their address space --
    // every per-lane twiddle load of the persistent kernels became a FLAT load, counted by lgkmcnt as well as vmcnt, so that each
    // re-deal's LDS wait also waited for the twiddles in flight; found in the ISA in round 6: forward +1.5-2 % at batch 1024, +4 % at
    // batch 4096 (15.2 M NTT/s = 0.498 of 8 TB/s), inverse +2 % at 4096; profiles/r06_ab_ntt_flat_loads.txt)
    w = pr.w; wp = pr.wp;
    return __hip_atomic_load(pr.viol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// Integer fallbacks of the fast-path kernels, kept out of line so that their register needs do not leak into the
// FP64 code (inlined, the allocator spilled ~100 VGPRs on the fast path).
// Round 6 measured the two alternatives the review asked for (profiles/r06_ab_ntt_redo.txt, tools/experiments/
// r06_ntt_redo_all_and_address_spaces.patch): (a) the fallback of EVERY N = 16384 persistent kernel in a k_ntt_redo_* launch of its own -- the
// fast kernels then need 92-94 VGPRs and no scratch at all (with the call: 128 VGPRs, 140-220 B of scratch around it) -- runs forward 13.2-13.3 /
// 15.0-15.1 M NTT/s at batch 1024 / 4096 and inverse 13.4-13.5 / 14.2 M against 13.35-13.4 / 15.2 M and 13.3-13.4 / 14.25 M with the call: the
// cleaner kernel gains what the extra (nearly always empty) dispatch costs, and a batch under tables that are not Shoup tables would run
// on 32 workgroups instead of the full grid -- the call stays; (b) address-space-qualified pointer parameters for these functions (their
// generic pointers make every access a FLAT operation, 70-145 per call) DOUBLE the caller's spills (k_ntt_inv_p: 24 -> 54 VGPRs, four of
// them inside the loop) and cost the inverse 3-9 % -- the generic pointers stay.
template <int LOGN, int LOGE>
__device__ __attribute__((noinline)) void slow_fwd(u64* px, u64* lds, const u64* roots, const u64* precon, u64 q) {
    using G = Geom<LOGN, LOGE>;
    const int tid = threadIdx.x;
    u64 v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = px[G::idxA(r, tid)];
    __syncthreads();                           // every wave holds its input before anyone overwrites it in place
    WgNtt<LOGN, LOGE>::forward_lazy(v, lds, tid, roots, precon, q);
    WgNtt<LOGN, LOGE>::final_reduce(v, q);
#pragma unroll
    for (int r = 0; r < G::E; ++r) px[G::idxB(r, tid)] = v[r];
}
template <int LOGN, int LOGE>
__device__ __attribute__((noinline)) void slow_inv(u64* px, u64* lds, const u64* iroots, const u64* iprecon, u64 q,
                                                   u64 inv_n, u64 inv_n_p, u64 inv_n_w, u64 inv_n_w_p) {
    using G = Geom<LOGN, LOGE>;
    const int tid = threadIdx.x;
    u64 v[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = px[G::idxB(r, tid)];
    __syncthreads();
    WgNtt<LOGN, LOGE>::template inverse<false, OrderedByCaller>(v, lds, tid, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);   // (the barrier above)
#pragma unroll
    for (int r = 0; r < G::E; ++r) px[G::idxA(r, tid)] = v[r];
}

// Tables that are NOT Shoup tables (benchmark/bench_fwd_ntt.cpp:36-42 feeds random words): the fast-path kernels find out from
// the prepare kernel's counter at kernel entry and run the integer butterflies (slow_fwd / slow_inv) on the whole batch -- correct,
// but through an out-of-line call whose twiddle loads are not scalar (measured 0.154 ms per 1024 polynomials against 0.119 ms for the
// dedicated integer kernels). So they also leave a HINT for the host: a tag of (table pointers, q, n) in a pinned host word. The
// next call with the same tag goes to the dedicated integer kernels straight away -- which are correct for ANY tables, so a stale
// or colliding hint can only cost speed -- still runs the prepare kernel, and clears the hint once the tables verify.
struct NttHint { unsigned long long* word; unsigned long long tag; };

// Round 4: what the FP64 fast path takes. The lazy kernels (q <= 2^51 (1 + 2^-7)) take words below min(1.25 q, 2^52) AS THEY ARE --
// the two-instruction conversion, no range reduction -- on the reduction schedule shifted by one stage (f64_arith.hpp
// lazy_fwd_reduce_after; the inverse's first stage only needs X + Y < 2.5 q and |X - Y| < 1.25 q): 64 of ~1500 VALU instructions per
// polynomial. Canonical inputs -- the case that matters -- are below q; words in [1.25 q, 4 q) (forward) / [1.25 q, 2 q) (inverse) are
// inside the Harvey contract, so those polynomials now go to the integer butterflies like out-of-contract ones (same answer, slower).
// The strict kernels (q up to 2^52) keep the full Harvey range and centre their inputs.
template <int LAZY>
__device__ __forceinline__ u64 fast_path_limit(u64 q, bool forward) {
    if constexpr (LAZY != 0) { const u64 l = q + (q >> 2); return l < (1ull << 52) ? l : (1ull << 52); }
    else return forward ? ((q << 2) < (1ull << 53) ? (q << 2) : (1ull << 53)) : ((q << 1) < (1ull << 53) ? (q << 1) : (1ull << 53));
}
// canonical result word of a fast-path transform. The strict kernels also serve moduli in [2^52, STRICT_NTT_MAX_Q) (f64_arith.hpp),
// whose residues need the conversion that does not assume 52 bits (`wide`, wave-uniform)
// (one wave-uniform branch around the whole store loop, not one per word)
template <int LAZY, class At>
__device__ __forceinline__ void fast_path_store(const double (&f)[1 << 4], u64* px, const Mod m, u64 q, At at) {
    if (LAZY == 0 && q >= (1ull << 52)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) px[at(r)] = hxf::from_f64_53(hxf::lift(f[r], m));
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) px[at(r)] = hxf::from_f64(hxf::lift(f[r], m));
    }
}
template <int LAZY, class At>
__device__ __forceinline__ void fast_path_store(const double (&f)[1 << 5], u64* px, const Mod m, u64 q, At at) {
    if (LAZY == 0 && q >= (1ull << 52)) {
#pragma unroll
        for (int r = 0; r < 32; ++r) px[at(r)] = hxf::from_f64_53(hxf::lift(f[r], m));
    } else {
#pragma unroll
        for (int r = 0; r < 32; ++r) px[at(r)] = hxf::from_f64(hxf::lift(f[r], m));
    }
}
template <int LAZY>
__device__ __forceinline__ double fast_path_input(u64 raw, const Mod m) {
    if constexpr (LAZY != 0) return hxf::to_f64_lt52(raw);
    else return hxf::reduce(hxf::to_f64(raw), m);
}

template <int LOGN, int LOGE, int LAZY, bool SEMI = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_fwd_x(u64* __restrict__ x, const u64* __restrict__ roots,
                                                                  const u64* __restrict__ precon, u64 q,
                                                                  const double* __restrict__ w,
                                                                  const double* __restrict__ wp,
                                                                  const u32* __restrict__ violations, u32 batch, NttHint hint) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const u32 p = blockIdx.x;
    if (p >= batch) return;
    u64* px = x + size_t(p) * G::N;
    // tables that are not genuine Shoup tables (benchmark/bench_fwd_ntt.cpp:36-42 feeds random ones) are known at kernel
    // entry and wave-uniform: straight to the integer butterflies, no FP64 transform first (round 4)
    if (*violations != 0) {
        if (p == 0 && tid == 0) *hint.word = hint.tag;
        slow_fwd<LOGN, LOGE>(px, lds, roots, precon, q);
        return;
    }
    const u64 limit = fast_path_limit<LAZY>(q, true);
    const Mod m{(double)q, 1.0 / (double)q};
    bool out_of_range = false;
    double f[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const u64 raw = px[G::idxA(r, tid)];
        out_of_range |= raw >= limit;
        f[r] = fast_path_input<LAZY>(raw, m);
    }
    // The FP64 transform runs unconditionally; whether its preconditions held for this polynomial is voted on
    // afterwards (a barrier at the very end costs nothing, one before the transform would put all 16 waves back
    // in lockstep). The input is still intact in memory for the integer fallback.
    WgNttF64<LOGN, LOGE, LAZY, 0, (LAZY != 0 ? 1 : 0), false, ntt_fwd_prio(LOGN), 0, SEMI, NTT_XSCHED ? 0 : -1>::template forward<true>(f, reinterpret_cast<double*>(lds), tid, w, wp, m);
    const bool slow = __syncthreads_or(out_of_range);
    if (!slow) {
        fast_path_store<LAZY>(f, px, m, q, [&](int r) { return G::idxB(r, tid); });
    } else {
        slow_fwd<LOGN, LOGE>(px, lds, roots, precon, q);
    }
}

template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_inv_x(u64* __restrict__ x, const u64* __restrict__ iroots,
                                                                  const u64* __restrict__ iprecon, u64 q, u64 inv_n,
                                                                  u64 inv_n_p, u64 inv_n_w, u64 inv_n_w_p,
                                                                  const double* __restrict__ w,
                                                                  const double* __restrict__ wp, hxf::InvScale sc,
                                                                  const u32* __restrict__ violations, u32 batch, NttHint hint) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const u32 p = blockIdx.x;
    if (p >= batch) return;
    u64* px = x + size_t(p) * G::N;
    if (*violations != 0) {                                                     // see k_ntt_fwd_x
        if (p == 0 && tid == 0) *hint.word = hint.tag;
        slow_inv<LOGN, LOGE>(px, lds, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
        return;
    }
    const u64 limit = fast_path_limit<LAZY>(q, false);
    const Mod m{(double)q, 1.0 / (double)q};
    bool out_of_range = false;
    double f[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const u64 raw = px[G::idxB(r, tid)];
        out_of_range |= raw >= limit;
        f[r] = fast_path_input<LAZY>(raw, m);
    }
    WgNttF64<LOGN, LOGE, LAZY, 0, 0, true, HX_FWD_PRIO, 0, false, -1, NTT_ISCHED != 0>::template inverse<true>(f, reinterpret_cast<double*>(lds), tid, w, wp, m, sc);   // no w/p table
    const bool slow = __syncthreads_or(out_of_range);                            // see k_ntt_fwd_x
    if (!slow) {
        fast_path_store<LAZY>(f, px, m, q, [&](int r) { return G::idxA(r, tid); });
    } else {
        slow_inv<LOGN, LOGE>(px, lds, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
    }
}

// "Did any thread of the workgroup see an out-of-range word?" without a barrier of its own. A __syncthreads_or behind
// the transform holds every wave until the slowest has finished, once per polynomial; the ballot is known BEFORE the
// transform, and every transform of a multi-wave workgroup contains a cross-wave re-deal with a barrier between its
// writes and reads (a one-wave workgroup executes its LDS operations in order): a flag set in LDS before the transform is
// therefore visible to every wave behind it. Three flags rotate so that the reset of a flag (by thread 0, behind the
// transform of round k: the flag of round k - 1, whose readers have all passed round k's barrier) is separated by round
// k + 1's barrier from its next setters in round k + 2. All waves take the same decision, and take it before any of
// them stores (the fallback re-reads the input, which is overwritten in place).
struct RangeVote {
    u32* flags;
    u32 k;
    static constexpr size_t BYTES = 16;
    __device__ __forceinline__ explicit RangeVote(char* at) : flags(reinterpret_cast<u32*>(at)), k(0) {
        if (threadIdx.x < 3) flags[threadIdx.x] = 0;
        __syncthreads();                                          // once per kernel
    }
    __device__ __forceinline__ void cast(bool bad) {
        if (bad) flags[k] = 1;
    }
    __device__ __forceinline__ bool result(int tid) {            // call behind the transform
        const bool bad = flags[k] != 0;
        const u32 prev = k == 0 ? 2 : k - 1;
        if (tid == 0) flags[prev] = 0;
        k = k == 2 ? 0 : k + 1;
        return bad;
    }
};

// Fast kernels WITHOUT a fallback of their own (ntt_inv_redo below, the N = 32768 half-transform kernels): a polynomial that fails the vote
// is only noted -- nothing of it has been stored -- and redone by the integer butterflies in a launch of its own behind the fast kernel
// (k_ntt_redo_*), which also takes the whole batch when the tables are not Shoup tables.
__device__ __forceinline__ void ntt_note_redo(const NttPrep& prep, u32 p, int tid) {
    if (tid == 0) prep.redo[1 + atomicAdd(prep.redo, 1u)] = p;
}
// Which inverse kernels do that: the strict-tier kernel at N = 16384. With the out-of-line fallback CALLED from it the allocator parks six of
// the sixteen prefetched words of the next polynomial in scratch -- a wait for them in the middle of the transform: 11.3 M against 12.0 M
// inverse NTT/s at q = 2^52 + 393217, batch 1024, 10.4 M against 11.9 M at batch 4096 with the fallback compiled out (round 5; the lazy
// kernels and the forward ones spill around the call only and measure the same either way, so they keep the call and save the dispatch).
template <int LOGN, int LAZY> constexpr bool ntt_inv_redo = (LOGN == 14 && LAZY == 0);

// Persistent variants (the default fast path): one workgroup per CU walks the batch, and the NEXT polynomial's words
// are requested into 2 E spare registers at the very start of the current transform. A lone 1024-thread workgroup per
// CU otherwise waits ~5 us for its 128 KiB input before every ~13 us transform (tools/ntt_timeline.hip) and pays the
// dispatch gap between workgroups. Vector memory returns in order, so the request must sit where nothing else is
// waited for until it has landed: the first two passes take their twiddles through the scalar cache (~14 k cycles).
// (Requested after the cross-wave re-deal instead, the first per-lane twiddle wait stalls on it:
// tools/experiments/persistent_prefetch.patch measured that 10 % SLOWER.)
template <int LOGN, int LOGE, int LAZY, bool SEMI = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_fwd_p(u64* __restrict__ x, const u64* __restrict__ roots,
                                                                  const u64* __restrict__ precon, u64 q, NttPrep prep,
                                                                  u32 batch, NttHint hint) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u64 limit = fast_path_limit<LAZY>(q, true);
    const Mod m{(double)q, 1.0 / (double)q};
    // the first polynomial is requested before anything is waited for
    u64 raw[G::E];
    {
        const int tid = threadIdx.x;
        const u64* p0 = x + size_t(blockIdx.x) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (p0 + G::idxA(r, 0))[u32(tid)];
    }
    const double *w, *wp;
    const bool bad_tables = ntt_tables_ready(prep, w, wp);   // counted by k_ntt_prepare
    if (bad_tables) {
        // Tables that are not genuine Shoup tables (benchmark/bench_fwd_ntt.cpp:36-42 feeds random ones): known at kernel entry,
        // the same for every polynomial and wave-uniform -- the whole batch goes straight through the integer butterflies.
        // (Rounds 2-3 ran the FP64 transform on every polynomial first and only then fell back: the transform twice.)
        if (blockIdx.x == 0 && threadIdx.x == 0) *hint.word = hint.tag;
#pragma unroll 1
        for (u32 p = blockIdx.x; p < batch; p += gridDim.x) {
            slow_fwd<LOGN, LOGE>(x + size_t(p) * G::N, lds, roots, precon, q);
            __syncthreads();
        }
        return;
    }
    RangeVote vote(reinterpret_cast<char*>(lds) + G::LDS_USED);
#pragma unroll 1
    for (u32 p = blockIdx.x; p < batch; p += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u64* px = x + size_t(p) * G::N;
        bool out_of_range = false;
        double f[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) {
            out_of_range |= raw[r] >= limit;
            f[r] = fast_path_input<LAZY>(raw[r], m);
        }
        vote.cast(out_of_range);
        const u32 pn = p + gridDim.x < batch ? p + gridDim.x : p;                // (last round: a harmless re-read)
        const u64* pnx = x + size_t(pn) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (pnx + G::idxA(r, 0))[u32(tid)];
        WgNttF64<LOGN, LOGE, LAZY, 0, (LAZY != 0 ? 1 : 0), false, ntt_fwd_prio(LOGN), 0, SEMI, NTT_XSCHED ? 0 : -1>::template forward<false>(f, reinterpret_cast<double*>(lds), tid, w, wp, m);
        const bool slow = vote.result(tid);                                      // see k_ntt_fwd_x, RangeVote
        if (!slow) {
            fast_path_store<LAZY>(f, px, m, q, [&](int r) { return G::idxB(r, tid); });
        } else {
            slow_fwd<LOGN, LOGE>(px, lds, roots, precon, q);
            __syncthreads();
        }
    }
}

template <int LOGN, int LOGE, int LAZY>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_inv_p(u64* __restrict__ x, const u64* __restrict__ iroots,
                                                                  const u64* __restrict__ iprecon, u64 q, u64 inv_n,
                                                                  u64 inv_n_p, u64 inv_n_w, u64 inv_n_w_p, NttPrep prep,
                                                                  hxf::InvScale sc, u32 batch, NttHint hint) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u64 limit = fast_path_limit<LAZY>(q, false);
    const Mod m{(double)q, 1.0 / (double)q};
    u64 raw[G::E];                                                // (requested before the wait for the tables: k_ntt_fwd_p)
    {
        const int tid = threadIdx.x;
        const u64* p0 = x + size_t(blockIdx.x) * G::N;
        const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (p0 + G::idxB(r, 0))[tB];
    }
    const double *w, *wp;
    const bool bad_tables = ntt_tables_ready(prep, w, wp);
    constexpr bool REDO = ntt_inv_redo<LOGN, LAZY>;
    if (bad_tables) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *hint.word = hint.tag;
        if constexpr (!REDO) {
#pragma unroll 1
            for (u32 p = blockIdx.x; p < batch; p += gridDim.x) {
                slow_inv<LOGN, LOGE>(x + size_t(p) * G::N, lds, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
                __syncthreads();
            }
        }
        return;
    }
    RangeVote vote(reinterpret_cast<char*>(lds) + G::LDS_USED);
    ReadersGate<G> gate(lds);                                     // inverse after inverse in one workgroup (ntt_core.hpp)
#pragma unroll 1
    for (u32 p = blockIdx.x; p < batch; p += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u64* px = x + size_t(p) * G::N;
        bool out_of_range = false;
        double f[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) {
            out_of_range |= raw[r] >= limit;
            f[r] = fast_path_input<LAZY>(raw[r], m);
        }
        vote.cast(out_of_range);
        // the inverse starts with its per-lane twiddle pass: the next input is requested behind that pass's twiddles,
        // i.e. after the first (wave-private) re-deal -- the following passes use the scalar cache
        const u32 pn = p + gridDim.x < batch ? p + gridDim.x : p;
        const u64* pnx = x + size_t(pn) * G::N;
        const u32 tB = u32(G::idxB(0, tid));
        auto request_next = [&] {
#pragma unroll
            for (int r = 0; r < G::E; ++r) raw[r] = (pnx + G::idxB(r, 0))[tB];
        };
        // (N = 32768 in one workgroup, HEXL_NTT_HALVES=0 only: the half-size exchanges bring barriers of their own except behind an inverse
        // transform's last round and in front of its first, wave-private, one -- the same inverse-after-inverse hole, closed with a barrier here)
        if constexpr (G::HALF_ONLY) __syncthreads();
        WgNttF64<LOGN, LOGE, LAZY, 0, 0, true, HX_FWD_PRIO, 0, false, -1, NTT_ISCHED != 0>::template inverse<false, decltype(request_next), (NTT_IPRE != 0), ReadersGate<G>>(f, reinterpret_cast<double*>(lds), tid, w, wp, m, sc, request_next, 0u, &gate);   // no w/p table
        const bool slow = vote.result(tid);
        if (!slow) {
            fast_path_store<LAZY>(f, px, m, q, [&](int r) { return G::idxA(r, tid); });
        } else if constexpr (REDO) {
            ntt_note_redo(prep, p, tid);
        } else {
            slow_inv<LOGN, LOGE>(px, lds, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
            __syncthreads();
        }
    }
}

// N = 32768 (beyond the reference's envelope): one polynomial = TWO 16384-point sub-transforms (round 5, the cut the slot-major keyswitch
// makes, keyswitch_x.hip k_ksh_*). 64 registers of polynomial per thread leave a 1024-thread workgroup no room for anything else, and
// 256 KiB do not fit the CU's LDS: k_ntt_fwd_x / k_ntt_inv_x<15, 5> exchange in half-size rounds, one workgroup per polynomial, no
// prefetch. Here the outermost stage is a radix-2 step ACROSS the two halves of the polynomial -- on load (forward: global stage 1, one
// twiddle) or behind the sub-transforms (inverse: the last stage, n^-1 folded in) -- and the other fourteen are two sub-transforms with
// the geometry, LDS footprint and register budget of the N = 16384 kernels (WgNttF64<14, 4, ..., TOP = 1>: stage numbers, reduction
// schedule and twiddle indices of the full transform). The transform is in place, so ONE persistent workgroup does both halves of a
// polynomial: the half that is not being transformed waits in the registers the N = 16384 kernels prefetch into.
// Same butterflies as the monolithic transform, canonical results: bit-identical.
// Polynomials that fail the range vote are only NOTED here (prep.redo: a counter and a list; nothing of theirs has been stored) and redone
// by the integer butterflies in a launch of their own right behind this one (k_ntt_redo_*), which also takes the whole batch when the
// tables are not Shoup tables. With the out-of-line fallback CALLED from these kernels -- inside the walk or behind it -- the allocator
// parks the second sub-transform's results in scratch on the fast path (inverse: 5.1 M against 6.2 M NTT/s).
template <int LAZY, bool SEMI = false>
__global__ __launch_bounds__(1024) void k_ntt_fwd_h(u64* __restrict__ x, const u64* __restrict__ roots, const u64* __restrict__ precon,
                                                    u64 q, NttPrep prep, u32 batch, NttHint hint) {
    using G = Geom<14, 4>;
    constexpr int FS = LAZY != 0 ? 1 : 0;
    using W = WgNttF64<14, 4, LAZY, 0, FS, false, ntt_fwd_prio(14), 1, SEMI, NTT_XSCHED ? 0 : -1>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u64 limit = fast_path_limit<LAZY>(q, true);
    const Mod m{(double)q, 1.0 / (double)q};
    const double *w, *wp;
    const bool bad_tables = ntt_tables_ready(prep, w, wp);
    if (bad_tables) {                                                           // see k_ntt_fwd_p; k_ntt_redo_fwd does the batch
        if (blockIdx.x == 0 && threadIdx.x == 0) *hint.word = hint.tag;
        return;
    }
    RangeVote vote(reinterpret_cast<char*>(lds) + G::LDS_USED);
    const double W1 = ((ctw_t)w)[1];                                            // global stage 1: one twiddle
    constexpr bool red = W::XS ? false : (LAZY == 0 || hxf::lazy_fwd_reduce_after(1, 15, LAZY ? LAZY : 3, FS));   // (X schedules: stage 1 is an N stage)
#pragma unroll 1
    for (u32 p = blockIdx.x; p < batch; p += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u64* px = x + size_t(p) * 2 * G::N;
        bool out_of_range = false;
        double u[G::E], v[G::E];
        {
            u64 lo[G::E], hi[G::E];
#pragma unroll
            for (int r = 0; r < G::E; ++r) lo[r] = (px + G::idxA(r, 0))[u32(tid)];
#pragma unroll
            for (int r = 0; r < G::E; ++r) hi[r] = (px + G::N + G::idxA(r, 0))[u32(tid)];
#pragma unroll
            for (int r = 0; r < G::E; ++r) {
                out_of_range |= (lo[r] >= limit) | (hi[r] >= limit);
                const double a = fast_path_input<LAZY>(lo[r], m);
                const double t = hxf::mul_mod(fast_path_input<LAZY>(hi[r], m), W1, m);
                u[r] = red ? hxf::reduce(a + t, m) : a + t;
                v[r] = red ? hxf::reduce(a - t, m) : a - t;
            }
        }
        vote.cast(out_of_range);
        W::template forward<false>(u, reinterpret_cast<double*>(lds), tid, w, wp, m, typename W::NoHook(), typename W::NoHook(), 0u);
        if (vote.result(tid)) {                                                  // before anything is stored
            ntt_note_redo(prep, p, tid);
            continue;
        }
        // (strict tier: the conversion that does not assume 52 bits for every modulus, see k_ntt_inv_h)
        auto store = [&](const double (&f)[G::E], u64* to) {
#pragma unroll
            for (int r = 0; r < G::E; ++r) to[G::idxB(r, tid)] = LAZY == 0 ? hxf::from_f64_53(hxf::lift(f[r], m)) : hxf::from_f64(hxf::lift(f[r], m));
        };
        store(u, px);
        W::template forward<false>(v, reinterpret_cast<double*>(lds), tid, w, wp, m, typename W::NoHook(), typename W::NoHook(), 1u);
        store(v, px + G::N);
    }
}

template <int LAZY>
__global__ __launch_bounds__(1024) void k_ntt_inv_h(u64* __restrict__ x, const u64* __restrict__ iroots, const u64* __restrict__ iprecon,
                                                    u64 q, NttPrep prep, hxf::InvScale sc, u32 batch, NttHint hint) {
    using G = Geom<14, 4>;
    using W = WgNttF64<14, 4, LAZY, 0, 0, true, HX_FWD_PRIO, 1>;      // no w/p table
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u64 limit = fast_path_limit<LAZY>(q, false);
    const Mod m{(double)q, 1.0 / (double)q};
    const double *w, *wp;
    const bool bad_tables = ntt_tables_ready(prep, w, wp);
    if (bad_tables) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *hint.word = hint.tag;
        return;
    }
    RangeVote vote(reinterpret_cast<char*>(lds) + G::LDS_USED);
    ReadersGate<G> gate(lds);                                     // inverse after inverse in one workgroup (ntt_core.hpp)
    // block 1 of a polynomial is requested inside the transform of block 0 (where k_ntt_inv_p requests its next polynomial), block 0 of the
    // NEXT polynomial behind the second transform, pair by pair between the stores; two votes per polynomial, nothing is stored before the
    // second
    u64 raw[G::E];
    {
        const u64* p0 = x + size_t(blockIdx.x) * 2 * G::N;
        const u32 tB = u32(G::idxB(0, threadIdx.x));
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (p0 + G::idxB(r, 0))[tB];
    }
#pragma unroll 1
    for (u32 p = blockIdx.x; p < batch; p += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u64* px = x + size_t(p) * 2 * G::N;
        const u32 pn = p + gridDim.x < batch ? p + gridDim.x : p;                // (last round: a harmless re-read)
        const u64* pnx = x + size_t(pn) * 2 * G::N;
        const u32 tB = u32(G::idxB(0, tid));
        auto request = [&](const u64* from) {
#pragma unroll
            for (int r = 0; r < G::E; ++r) raw[r] = (from + G::idxB(r, 0))[tB];
        };
        bool out_of_range = false;
        double u[G::E], v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) {
            out_of_range |= raw[r] >= limit;
            u[r] = fast_path_input<LAZY>(raw[r], m);
        }
        vote.cast(out_of_range);
        auto request_other_half = [&] { request(px + G::N); };
        W::template inverse<false, decltype(request_other_half), false, ReadersGate<G>>(u, reinterpret_cast<double*>(lds), tid, w, wp, m, sc, request_other_half, 0u, &gate);
        bool slow = vote.result(tid);
        if (!slow) {
            out_of_range = false;
#pragma unroll
            for (int r = 0; r < G::E; ++r) {
                out_of_range |= raw[r] >= limit;
                v[r] = fast_path_input<LAZY>(raw[r], m);
            }
            vote.cast(out_of_range);
            W::template inverse<false, typename W::NoHook, false, ReadersGate<G>>(v, reinterpret_cast<double*>(lds), tid, w, wp, m, sc, typename W::NoHook(), 1u, &gate);
            slow = vote.result(tid);
        }
        if (slow) {
            ntt_note_redo(prep, p, tid);
            request(pnx);
            continue;
        }
        // the inverse's last stage across the halves, n^-1 folded in (inv_stages_f64's fused stage on register pairs), pair by pair: a pair
        // is stored and its registers take the next polynomial's word -- all sixteen requests up front, beside 64 live registers of
        // results, had the compiler park the arriving words in scratch one by one (a wait per word)
        // (strict tier: the conversion that does not assume 52 bits for every modulus -- a wave-uniform choice between two copies of this
        // loop, as in fast_path_store, had the allocator park the results in scratch in front of the branch)
#pragma unroll
        for (int r = 0; r < G::E; ++r) {
            double pr2[2] = {u[r], v[r]};
            inv_stages_f64<2, 0, 1, 14, 15, true, 3, true, true>(pr2, 0u, w, wp, m, sc);
            const double c0 = hxf::lift(pr2[0], m), c1 = hxf::lift(pr2[1], m);
            (px + G::idxA(r, 0))[u32(tid)] = LAZY == 0 ? hxf::from_f64_53(c0) : hxf::from_f64(c0);
            (px + G::N + G::idxA(r, 0))[u32(tid)] = LAZY == 0 ? hxf::from_f64_53(c1) : hxf::from_f64(c1);
            raw[r] = (pnx + G::idxB(r, 0))[tB];
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// the integer butterflies for what a fast kernel without a fallback of its own left (the half-transform kernels above, k_ntt_inv_p<14, 4, 0>):
// the noted polynomials, or the whole batch under tables that are not Shoup tables
template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_redo_fwd(u64* __restrict__ x, const u64* __restrict__ roots, const u64* __restrict__ precon,
                                                                     u64 q, NttPrep prep, u32 batch) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    // (both words requested before either is looked at: the launch is nearly always empty and its duration is these loads' latency)
    const u32 bad = __hip_atomic_load(prep.viol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 noted = __hip_atomic_load(prep.redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool all = bad != 0;
    const u32 n = all ? batch : noted;
#pragma unroll 1
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u32 p = all ? i : prep.redo[1 + i];
        slow_fwd<LOGN, LOGE>(x + (size_t(p) << LOGN), lds, roots, precon, q);
        __syncthreads();
    }
}
template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_redo_inv(u64* __restrict__ x, const u64* __restrict__ iroots, const u64* __restrict__ iprecon,
                                                                     u64 q, u64 inv_n, u64 inv_n_p, u64 inv_n_w, u64 inv_n_w_p, NttPrep prep, u32 batch) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    // (both words requested before either is looked at: the launch is nearly always empty and its duration is these loads' latency)
    const u32 bad = __hip_atomic_load(prep.viol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 noted = __hip_atomic_load(prep.redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool all = bad != 0;
    const u32 n = all ? batch : noted;
#pragma unroll 1
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u32 p = all ? i : prep.redo[1 + i];
        slow_inv<LOGN, LOGE>(x + (size_t(p) << LOGN), lds, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
        __syncthreads();
    }
}

// HEXL_NTT_HALVES=0: N = 32768 back on the monolithic half-size-exchange kernels (A/B)
static bool halves_enabled() {
    static const bool v = [] { const char* e = getenv("HEXL_NTT_HALVES"); return !(e && atoi(e) == 0); }();
    return v;
}

static bool fast_path_enabled() {
    static const bool v = [] { const char* e = getenv("HEXL_NTT_INT"); return !(e && atoi(e) == 1); }();
    return v;
}

static int ensure_ntt_hint(hexl_ctx* ctx) {                        // NttHint: four pinned, device-visible words
    if (!ctx->h_ntt_hint) {
        HX_CHECK(hipHostMalloc((void**)&ctx->h_ntt_hint, 4 * sizeof(unsigned long long), hipHostMallocMapped));
        memset(ctx->h_ntt_hint, 0, 4 * sizeof(unsigned long long));
        HX_CHECK(hipHostGetDevicePointer((void**)&ctx->d_ntt_hint, ctx->h_ntt_hint, 0));
    }
    return 0;
}

// device scratch for the derived tables: a ring of 64 violation counters (one per launch, round robin), then [w | w/p] (n doubles each)
static int reserve_tables(hexl_ctx* ctx, u64 n, NttPrep* pr) {
    constexpr size_t HEAD = 256;                                                        // 64 violation counters
    const size_t bytes = HEAD + 2 * n * sizeof(double);
    const void* before = ctx->d_ntt_tab;
    int rc = hx_reserve_device(ctx, &ctx->d_ntt_tab, &ctx->d_ntt_tab_bytes, bytes);
    if (rc) return rc;
    if (ctx->d_ntt_tab != before) HX_CHECK(hipMemsetAsync(ctx->d_ntt_tab, 0, HEAD, ctx->stream));
    u32* counters = (u32*)ctx->d_ntt_tab;
    pr->w = (double*)((char*)ctx->d_ntt_tab + HEAD);
    pr->wp = pr->w + n;
    if (int rch = ensure_ntt_hint(ctx)) return rch;
    const u32 seq = ctx->ntt_seq++;
    pr->viol = counters + (seq & 63);
    pr->viol_later = counters + ((seq + 32) & 63);
    pr->n = (u32)n;
    pr->redo = nullptr;
    return 0;
}
// the launch behind the fast kernel is nearly always EMPTY: 32 workgroups (an empty dispatch of one 141 KiB workgroup per CU measured 4.7 us,
// 5 % of a 512-polynomial launch); a batch under tables that are not Shoup tables runs there once, then the host hint sends it to the
// dedicated integer kernels
static unsigned redo_grid(unsigned fast_grid) { return fast_grid < 32u ? fast_grid : 32u; }
static int reserve_redo(hexl_ctx* ctx, size_t batch, NttPrep* pr) {
    int rc = hx_reserve_device(ctx, &ctx->d_ntt_redo, &ctx->d_ntt_redo_bytes, (batch + 1) * sizeof(u32));
    pr->redo = (u32*)ctx->d_ntt_redo;
    return rc;
}
// the preparation as a launch of its own, in front of a transform kernel that does not do it itself
static int launch_prepare(hexl_ctx* ctx, const u64* roots, const u64* precon, u64 q, const NttPrep& pr) {
    hipLaunchKernelGGL(k_ntt_prepare, dim3((pr.n + 255) / 256), dim3(256), 0, ctx->stream, roots, precon, q, pr);
    return (int)hipGetLastError();
}
template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_fwd(u64* __restrict__ x,
                                                                const u64* __restrict__ roots,
                                                                const u64* __restrict__ precon, u64 q,
                                                                u32 batch, const u32* __restrict__ viol = nullptr,
                                                                unsigned long long* hint_word = nullptr) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    // hinted route (NttHint above): the tables verified this time -> the next call takes the fast path again
    if (hint_word && blockIdx.x == 0 && tid == 0 && *viol == 0) *hint_word = 0;
    {
        const u32 p = blockIdx.x;   // one workgroup per polynomial (grid == batch)
        if (p >= batch) return;
        u64* px = x + size_t(p) * G::N;
        u64 v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = px[G::idxA(r, tid)];
        WgNtt<LOGN, LOGE>::template forward_lazy<true>(v, lds, tid, roots, precon, q);
        WgNtt<LOGN, LOGE>::final_reduce(v, q);
#pragma unroll
        for (int r = 0; r < G::E; ++r) px[G::idxB(r, tid)] = v[r];
    }
}

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_inv(u64* __restrict__ x,
                                                                const u64* __restrict__ iroots,
                                                                const u64* __restrict__ iprecon, u64 q,
                                                                u64 inv_n, u64 inv_n_p, u64 inv_n_w,
                                                                u64 inv_n_w_p, u32 batch, const u32* __restrict__ viol = nullptr,
                                                                unsigned long long* hint_word = nullptr) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    if (hint_word && blockIdx.x == 0 && tid == 0 && *viol == 0) *hint_word = 0;      // see k_ntt_fwd
    {
        const u32 p = blockIdx.x;   // one workgroup per polynomial (grid == batch)
        if (p >= batch) return;
        u64* px = x + size_t(p) * G::N;
        u64 v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = px[G::idxB(r, tid)];
        WgNtt<LOGN, LOGE>::template inverse<true>(v, lds, tid, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
#pragma unroll
        for (int r = 0; r < G::E; ++r) px[G::idxA(r, tid)] = v[r];
    }
}

// Persistent integer inverse (moduli >= 2^52, or HEXL_NTT_INT=1) for N = 16384: one workgroup per CU walks the batch with
// the next polynomial's words already requested, like k_ntt_inv_p: 8.3 M against 7.6 M inverse NTT/s. The same for the
// forward transform measured SLOWER (8.1 M against 8.6 M; N = 1024: 165 M against 195 M) -- the integer butterflies are
// bound by their instruction count, the prefetch registers cost more than the hidden load latency gains -- and is not kept.
template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_inv_ip(u64* __restrict__ x, const u64* __restrict__ iroots,
                                                                   const u64* __restrict__ iprecon, u64 q, u64 inv_n,
                                                                   u64 inv_n_p, u64 inv_n_w, u64 inv_n_w_p, u32 batch,
                                                                   const u32* __restrict__ viol = nullptr,
                                                                   unsigned long long* hint_word = nullptr) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    if (hint_word && blockIdx.x == 0 && threadIdx.x == 0 && *viol == 0) *hint_word = 0;      // see k_ntt_fwd
    u64 raw[G::E];
    {
        const int tid = threadIdx.x;
        const u64* p0 = x + size_t(blockIdx.x) * G::N;
        const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (p0 + G::idxB(r, 0))[tB];
    }
    ReadersGate<G> gate(lds);                                     // inverse after inverse in one workgroup (ntt_core.hpp)
#pragma unroll 1
    for (u32 p = blockIdx.x; p < batch; p += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u64* px = x + size_t(p) * G::N;
        u64 v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = raw[r];
        const u32 pn = p + gridDim.x < batch ? p + gridDim.x : p;
        const u64* pnx = x + size_t(pn) * G::N;
        const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (pnx + G::idxB(r, 0))[tB];
        const u64* tw = iroots + opaque_zero();
        WgNtt<LOGN, LOGE>::template inverse<false, ReadersGate<G>>(v, lds, tid, tw, iprecon + (tw - iroots), q, inv_n, inv_n_p, inv_n_w, inv_n_w_p, &gate);
#pragma unroll
        for (int r = 0; r < G::E; ++r) (px + G::idxA(r, 0))[u32(tid)] = v[r];
    }
}

u32 hx_loge_for(u32 logn) {
    // N = 16384: 16 coefficients per thread x 1024 threads (4 waves/SIMD) measured ~10 % faster than 32 x 512
    return (logn == 14 || logn <= 10) ? 4 : 5;
}

u32 hx_idxB(u32 logn, u32 r, u32 tid) {
    const u32 loge = hx_loge_for(logn);
    const u32 P = (logn + loge - 1) / loge, KL = logn - (P - 1) * loge;
    const u32 WB = logn - loge < 6 ? logn - loge : 6;          // mirrors Geom::idxB (ntt_core.hpp)
    const u32 grp = ((tid >> WB) << (loge - KL + WB)) + ((r >> KL) << WB) + (tid & ((1u << WB) - 1));
    return (grp << KL) + (r & ((1u << KL) - 1));
}

template <int LOGN, int LOGE>
static int launch_fwd(hexl_ctx* ctx, u64* x, size_t batch, const u64* roots, const u64* precon, u64 q) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc = once.run(ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_fwd<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED));
            return 0;
        }))
        return rc;
    // (a persistent forward integer kernel measured no faster in rounds 2 and 4 -- the integer butterflies are bound by their instruction
    // count, the prefetch registers cost what the hidden load latency gains -- and is gone)
    hipLaunchKernelGGL((k_ntt_fwd<LOGN, LOGE>), dim3((unsigned)batch), dim3(G::T), G::LDS_USED, ctx->stream, x,
                       roots, precon, q, (u32)batch, ctx->ntt_clear_viol, ctx->ntt_clear_viol ? ctx->ntt_hint_word : nullptr);
    return (int)hipGetLastError();
}

template <int LOGN, int LOGE>
static int launch_inv(hexl_ctx* ctx, u64* x, size_t batch, const u64* ir, const u64* ip, u64 q, u64 a, u64 ap,
                      u64 b, u64 bp) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc = once.run(ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_inv<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED));
            return 0;
        }))
        return rc;
    static const int persist = [] { const char* e = getenv("HEXL_NTT_PERSIST"); return e ? atoi(e) : 1; }();
    const size_t slots = size_t(ctx->num_cu) * (G::LDS_USED > 80 * 1024 ? 1 : (160 * 1024) / G::LDS_USED);
    if constexpr (LOGN == 14 && LOGE == 4) if (persist && batch > slots) {
        static PerDeviceOnce once_p;
        if (int rc = once_p.run(ctx->device, [] {
                HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_inv_ip<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED));
                return 0;
            }))
            return rc;
        hipLaunchKernelGGL((k_ntt_inv_ip<LOGN, LOGE>), dim3((unsigned)slots), dim3(G::T), G::LDS_USED, ctx->stream, x, ir, ip,
                           q, a, ap, b, bp, (u32)batch, ctx->ntt_clear_viol, ctx->ntt_clear_viol ? ctx->ntt_hint_word : nullptr);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((k_ntt_inv<LOGN, LOGE>), dim3((unsigned)batch), dim3(G::T), G::LDS_USED, ctx->stream, x, ir,
                       ip, q, a, ap, b, bp, (u32)batch, ctx->ntt_clear_viol, ctx->ntt_clear_viol ? ctx->ntt_hint_word : nullptr);
    return (int)hipGetLastError();
}

template <int LOGN, int LOGE, int LAZY, bool SEMI = false>
static int launch_fwd_x(hexl_ctx* ctx, u64* x, size_t batch, const u64* roots, const u64* precon, u64 q, NttPrep pr) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc = once.run(ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_fwd_x<LOGN, LOGE, LAZY, SEMI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED));
            return 0;
        }))
        return rc;
    // persistent workgroups with input prefetch wherever 2 E more registers fit (not N = 32768: 64 data registers at
    // 1024 threads) and the batch fills the chip more than once; HEXL_NTT_PERSIST=0 keeps one workgroup per polynomial
    static const int persist = [] { const char* e = getenv("HEXL_NTT_PERSIST"); return e ? atoi(e) : 1; }();
    const size_t slots = size_t(ctx->num_cu) * (G::LDS_USED > 80 * 1024 ? 1 : (160 * 1024) / G::LDS_USED);
    if (persist && !G::HALF_ONLY && batch > slots) {
        static PerDeviceOnce once_p;
        if (int rc = once_p.run(ctx->device, [] {
                HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_fwd_p<LOGN, LOGE, LAZY, SEMI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G::LDS_USED + RangeVote::BYTES)));
                return 0;
            }))
            return rc;
        if (int rc = launch_prepare(ctx, roots, precon, q, pr)) return rc;
        hipLaunchKernelGGL((k_ntt_fwd_p<LOGN, LOGE, LAZY, SEMI>), dim3((unsigned)slots), dim3(G::T), G::LDS_USED + RangeVote::BYTES, ctx->stream, x,
                           roots, precon, q, pr, (u32)batch, NttHint{ctx->ntt_hint_word, ctx->ntt_hint_tag});
        return (int)hipGetLastError();
    }
    if (int rc = launch_prepare(ctx, roots, precon, q, pr)) return rc;
    hipLaunchKernelGGL((k_ntt_fwd_x<LOGN, LOGE, LAZY, SEMI>), dim3((unsigned)batch), dim3(G::T), G::LDS_USED, ctx->stream, x,
                       roots, precon, q, (const double*)pr.w, (const double*)pr.wp, (const u32*)pr.viol, (u32)batch, NttHint{ctx->ntt_hint_word, ctx->ntt_hint_tag});
    return (int)hipGetLastError();
}

template <int LOGN, int LOGE, int LAZY>
static int launch_inv_x(hexl_ctx* ctx, u64* x, size_t batch, const u64* ir, const u64* ip, u64 q, u64 a, u64 ap, u64 b,
                        u64 bp, NttPrep pr, hxf::InvScale sc) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc = once.run(ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_inv_x<LOGN, LOGE, LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED));
            return 0;
        }))
        return rc;
    static const int persist = [] { const char* e = getenv("HEXL_NTT_PERSIST"); return e ? atoi(e) : 1; }();
    const size_t slots = size_t(ctx->num_cu) * (G::LDS_USED > 80 * 1024 ? 1 : (160 * 1024) / G::LDS_USED);
    if (persist && !G::HALF_ONLY && batch > slots) {
        static PerDeviceOnce once_p;
        if (int rc = once_p.run(ctx->device, [] {
                HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_inv_p<LOGN, LOGE, LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G::LDS_USED + RangeVote::BYTES)));
                return 0;
            }))
            return rc;
        if constexpr (ntt_inv_redo<LOGN, LAZY>) {                                // fallback in a launch of its own behind the fast kernel
            static PerDeviceOnce once_r;
            if (int rc = once_r.run(ctx->device, [] {
                    HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_redo_inv<LOGN, LOGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_USED));
                    return 0;
                }))
                return rc;
            if (int rc = reserve_redo(ctx, batch, &pr)) return rc;               // (the preparation zeroes the counter)
        }
        if (int rc = launch_prepare(ctx, ir, ip, q, pr)) return rc;
        hipLaunchKernelGGL((k_ntt_inv_p<LOGN, LOGE, LAZY>), dim3((unsigned)slots), dim3(G::T), G::LDS_USED + RangeVote::BYTES, ctx->stream, x,
                           ir, ip, q, a, ap, b, bp, pr, sc, (u32)batch, NttHint{ctx->ntt_hint_word, ctx->ntt_hint_tag});
        if constexpr (ntt_inv_redo<LOGN, LAZY>)
            hipLaunchKernelGGL((k_ntt_redo_inv<LOGN, LOGE>), dim3(redo_grid((unsigned)slots)), dim3(G::T), G::LDS_USED, ctx->stream, x, ir, ip, q, a, ap, b, bp,
                               pr, (u32)batch);
        return (int)hipGetLastError();
    }
    if (int rc = launch_prepare(ctx, ir, ip, q, pr)) return rc;
    hipLaunchKernelGGL((k_ntt_inv_x<LOGN, LOGE, LAZY>), dim3((unsigned)batch), dim3(G::T), G::LDS_USED, ctx->stream, x,
                       ir, ip, q, a, ap, b, bp, (const double*)pr.w, (const double*)pr.wp, sc, (const u32*)pr.viol, (u32)batch, NttHint{ctx->ntt_hint_word, ctx->ntt_hint_tag});
    return (int)hipGetLastError();
}

template <int LAZY, bool SEMI = false>
static int launch_fwd_h(hexl_ctx* ctx, u64* x, size_t batch, const u64* roots, const u64* precon, u64 q, NttPrep pr) {
    using G = Geom<14, 4>;
    constexpr size_t LDS = G::LDS_USED + RangeVote::BYTES, LDS_INT = Geom<15, 5>::LDS_USED;
    static PerDeviceOnce once;
    if (int rc = once.run(ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_fwd_h<LAZY, SEMI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_redo_fwd<15, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_INT));
            return 0;
        }))
        return rc;
    if (int rc = reserve_redo(ctx, batch, &pr)) return rc;
    if (int rc = launch_prepare(ctx, roots, precon, q, pr)) return rc;           // (also zeroes the redo counter)
    const unsigned grid = (unsigned)(batch < (size_t)ctx->num_cu ? batch : (size_t)ctx->num_cu);
    hipLaunchKernelGGL((k_ntt_fwd_h<LAZY, SEMI>), dim3(grid), dim3(G::T), LDS, ctx->stream, x, roots, precon, q, pr, (u32)batch,
                       NttHint{ctx->ntt_hint_word, ctx->ntt_hint_tag});
    hipLaunchKernelGGL((k_ntt_redo_fwd<15, 5>), dim3(redo_grid(grid)), dim3(G::T), LDS_INT, ctx->stream, x, roots, precon, q, pr, (u32)batch);
    return (int)hipGetLastError();
}
template <int LAZY>
static int launch_inv_h(hexl_ctx* ctx, u64* x, size_t batch, const u64* ir, const u64* ip, u64 q, u64 a, u64 ap, u64 b, u64 bp, NttPrep pr,
                        hxf::InvScale sc) {
    using G = Geom<14, 4>;
    constexpr size_t LDS = G::LDS_USED + RangeVote::BYTES, LDS_INT = Geom<15, 5>::LDS_USED;
    static PerDeviceOnce once;
    if (int rc = once.run(ctx->device, [] {
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_inv_h<LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_redo_inv<15, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_INT));
            return 0;
        }))
        return rc;
    if (int rc = reserve_redo(ctx, batch, &pr)) return rc;
    if (int rc = launch_prepare(ctx, ir, ip, q, pr)) return rc;
    const unsigned grid = (unsigned)(batch < (size_t)ctx->num_cu ? batch : (size_t)ctx->num_cu);
    hipLaunchKernelGGL((k_ntt_inv_h<LAZY>), dim3(grid), dim3(G::T), LDS, ctx->stream, x, ir, ip, q, pr, sc, (u32)batch,
                       NttHint{ctx->ntt_hint_word, ctx->ntt_hint_tag});
    hipLaunchKernelGGL((k_ntt_redo_inv<15, 5>), dim3(redo_grid(grid)), dim3(G::T), LDS_INT, ctx->stream, x, ir, ip, q, a, ap, b, bp, pr, (u32)batch);
    return (int)hipGetLastError();
}

// FP64 transforms of N = 2048 .. 8192: 16 coefficients per thread where that measured faster than 32 (twice the waves
// per CU, but a different run length per lane in the B-order access): forward N = 2048 (+17 %) and 8192 (+16 %), inverse
// N = 2048 (+12 %); N = 4096 lost 16 % both ways, the inverse at 8192 11 % (tools/ntt_n_sweep.py).
// HEXL_NTT_E16 = bit mask over logn - 11 forces the choice for both directions.
static bool small_e16(int logn, bool fwd) {
    static const int mask = [] { const char* e = getenv("HEXL_NTT_E16"); return e ? atoi(e) : -1; }();
    const int m = mask >= 0 ? mask : (fwd ? 0b101 : 0b001);
    return (m >> (logn - 11)) & 1;
}

template <int LAZY, bool SEMI = false>
static int dispatch_fwd_x(int logn, hexl_ctx* c, u64* x, size_t batch, const u64* r, const u64* p, u64 q, const NttPrep& pr) {
    switch (logn) {
        case 10: return launch_fwd_x<10, 4, LAZY, SEMI>(c, x, batch, r, p, q, pr);
        case 11: return small_e16(11, true) ? launch_fwd_x<11, 4, LAZY, SEMI>(c, x, batch, r, p, q, pr)
                                   : launch_fwd_x<11, 5, LAZY, SEMI>(c, x, batch, r, p, q, pr);
        case 12: return small_e16(12, true) ? launch_fwd_x<12, 4, LAZY, SEMI>(c, x, batch, r, p, q, pr)
                                   : launch_fwd_x<12, 5, LAZY, SEMI>(c, x, batch, r, p, q, pr);
        case 13: return small_e16(13, true) ? launch_fwd_x<13, 4, LAZY, SEMI>(c, x, batch, r, p, q, pr)
                                   : launch_fwd_x<13, 5, LAZY, SEMI>(c, x, batch, r, p, q, pr);
        case 14: return launch_fwd_x<14, 4, LAZY, SEMI>(c, x, batch, r, p, q, pr);
        case 15: return halves_enabled() ? launch_fwd_h<LAZY, SEMI>(c, x, batch, r, p, q, pr)       // beyond the reference: two sub-transforms
                                         : launch_fwd_x<15, 5, LAZY, SEMI>(c, x, batch, r, p, q, pr);   // (half-size exchanges)
        default: return HEXL_E_BADARG;
    }
}
template <int LAZY>
static int dispatch_inv_x(int logn, hexl_ctx* c, u64* x, size_t batch, const u64* r, const u64* p, u64 q, u64 a, u64 ap,
                          u64 b, u64 bp, const NttPrep& pr, hxf::InvScale sc) {
    switch (logn) {
        case 10: return launch_inv_x<10, 4, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc);
        case 11: return small_e16(11, false) ? launch_inv_x<11, 4, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc)
                                   : launch_inv_x<11, 5, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc);
        case 12: return small_e16(12, false) ? launch_inv_x<12, 4, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc)
                                   : launch_inv_x<12, 5, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc);
        case 13: return small_e16(13, false) ? launch_inv_x<13, 4, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc)
                                   : launch_inv_x<13, 5, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc);
        case 14: return launch_inv_x<14, 4, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc);
        case 15: return halves_enabled() ? launch_inv_h<LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc)
                                         : launch_inv_x<15, 5, LAZY>(c, x, batch, r, p, q, a, ap, b, bp, pr, sc);
        default: return HEXL_E_BADARG;
    }
}

static int ilog2_exact(u64 n) {
    for (int l = 0; l < 63; ++l)
        if ((1ULL << l) == n) return l;
    return -1;
}

// tag of a table set (never 0) and its hint slot; true when a previous launch on this context flagged exactly this set
static bool ntt_hinted(hexl_ctx* ctx, const u64* t0, const u64* t1, u64 q, u64 n, int dir) {
    u64 z = (u64)(uintptr_t)t0 * 0x9E3779B97F4A7C15ull ^ (u64)(uintptr_t)t1 * 0xBF58476D1CE4E5B9ull ^ q * 0x94D049BB133111EBull ^ (n << 1) ^ (u64)dir;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    ctx->ntt_hint_tag = z | 1;
    const u32 slot = (u32)(z >> 60) & 3;
    ctx->ntt_hint_word = ctx->d_ntt_hint + slot;
    return ((volatile unsigned long long*)ctx->h_ntt_hint)[slot] == ctx->ntt_hint_tag;
}

int hx_launch_ntt_fwd(hexl_ctx* ctx, u64* x, size_t batch, const u64* roots, const u64* precon, u64 q, u64 n) {
    if (!batch) return 0;
    const int logn = ilog2_exact(n);
    ctx->ntt_clear_viol = nullptr;
    if (fast_path_enabled() && q >= (1ull << 16) && q < hxf::STRICT_NTT_MAX_Q) {
        NttPrep pr;
        // a previous call found these tables not to be Shoup tables: the dedicated integer kernels, which also clear the hint
        // once the prepare kernel (still run for a hinted set) finds the tables genuine
        if (int rch = ensure_ntt_hint(ctx)) return rch;
        const bool hinted = ntt_hinted(ctx, roots, precon, q, n, 0);
        int rc = reserve_tables(ctx, n, &pr);
        if (!rc && hinted) rc = launch_prepare(ctx, roots, precon, q, pr);      // (the fast-path launchers below prepare for themselves)
        if (rc) return rc;
        if (hinted) ctx->ntt_clear_viol = pr.viol;
        const int period = hxf::lazy_period_for((double)q);       // fewer range reductions for smaller moduli (N = 16384)
        if (!hinted) {
            if (logn == 14 && period == 12) return launch_fwd_x<14, 4, 12>(ctx, x, batch, roots, precon, q, pr);
            if (logn == 14 && period == 6) return launch_fwd_x<14, 4, 6>(ctx, x, batch, roots, precon, q, pr);
            if (period) return dispatch_fwd_x<3>(logn, ctx, x, batch, roots, precon, q, pr);
            // strict tier: semi-strict butterflies (f64_arith.hpp ct_bfly_semi, 11 instead of 14 instructions) in the wave-uniform passes up to
            // 2^52 (1 + 2^-20) -- SURVEY 8d's q = 2^52 + 393217 included --, the plain strict ones above
            return (double)q <= hxf::SEMI_MAX_MODULUS ? dispatch_fwd_x<0, true>(logn, ctx, x, batch, roots, precon, q, pr)
                                                                        : dispatch_fwd_x<0>(logn, ctx, x, batch, roots, precon, q, pr);
        }
    }
    switch (logn) {
        case 10: return launch_fwd<10, 4>(ctx, x, batch, roots, precon, q);
        case 11: return launch_fwd<11, 5>(ctx, x, batch, roots, precon, q);
        case 12: return launch_fwd<12, 5>(ctx, x, batch, roots, precon, q);
        case 13: return launch_fwd<13, 5>(ctx, x, batch, roots, precon, q);
        case 14: return launch_fwd<14, 4>(ctx, x, batch, roots, precon, q);
        case 15: return launch_fwd<15, 5>(ctx, x, batch, roots, precon, q);
        default: return HEXL_E_BADARG;
    }
}

int hx_launch_ntt_inv(hexl_ctx* ctx, u64* x, size_t batch, const u64* ir, const u64* ip, u64 q, u64 a, u64 ap,
                      u64 b, u64 bp, u64 n) {
    if (!batch) return 0;
    const int logn = ilog2_exact(n);
    ctx->ntt_clear_viol = nullptr;
    if (fast_path_enabled() && q >= (1ull << 16) && q < hxf::STRICT_NTT_MAX_Q && a < q && b < q) {
        NttPrep pr;
        if (int rch = ensure_ntt_hint(ctx)) return rch;
        const bool hinted = ntt_hinted(ctx, ir, ip, q, n, 1);     // see hx_launch_ntt_fwd
        int rc = reserve_tables(ctx, n, &pr);
        if (!rc && hinted) rc = launch_prepare(ctx, ir, ip, q, pr);
        if (rc) return rc;
        if (hinted) ctx->ntt_clear_viol = pr.viol;
        const double pd = (double)q;
        auto centre = [&](u64 v) { return v > q / 2 ? (double)v - pd : (double)v; };
        hxf::InvScale sc;
        sc.n = centre(a); sc.n_p = sc.n / pd; sc.nw = centre(b); sc.nw_p = sc.nw / pd;
        if (!hinted)
            return pd <= hxf::LAZY_MAX_MODULUS ? dispatch_inv_x<3>(logn, ctx, x, batch, ir, ip, q, a, ap, b, bp, pr, sc)
                                               : dispatch_inv_x<0>(logn, ctx, x, batch, ir, ip, q, a, ap, b, bp, pr, sc);
    }
    switch (logn) {
        case 10: return launch_inv<10, 4>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 11: return launch_inv<11, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 12: return launch_inv<12, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 13: return launch_inv<13, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 14: return launch_inv<14, 4>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 15: return launch_inv<15, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        default: return HEXL_E_BADARG;
    }
}
