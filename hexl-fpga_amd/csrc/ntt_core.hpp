// ntt_core.hpp -- workgroup-level negacyclic NTT / INTT for gfx950 (CDNA4), 64-bit Harvey/Shoup
// lazy butterflies, op-for-op identical to the reference FPGA kernels:
//   forward : device/fwd_ntt.cpp:289-385   (== tests/test_utils/ntt.cpp:474-548)
//   inverse : device/inv_ntt.cpp:286-437   (== tests/test_utils/ntt.cpp:580-659)
//
// MI355X mapping (one workgroup == one polynomial):
//   * N = 2^LOGN coefficients live in registers, E = 2^LOGE per thread, T = N/E threads.
//   * A transform is P = ceil(LOGN/LOGE) register passes of up to LOGE radix-2 stages each
//     (a radix-E butterfly network with static register indices); between passes the
//     polynomial is re-dealt to the threads through LDS (N*8 B + padding <= 160 KiB/CU).
//   * "A layout"  idx = r*T + tid          : wave reads/writes 512-B contiguous runs of HBM
//     "B layout"  idx = hi(r)*.. + tid*2^KL + lo(r): each lane owns 2^KL contiguous words.
//     forward goes A -> B, inverse goes B -> A, so INTT(NTT(x)) needs no data movement and
//     every intermediate a caller never sees can stay in "B order" ([r][tid], coalesced).
//   * twiddles come from the caller's tables (bit-reversed order) through L2; the first
//     pass's indices are uniform so the compiler emits scalar loads for them.
#pragma once
#ifndef HX_IFWD_PRIO
#define HX_IFWD_PRIO 0   // four digits, pass k of an integer forward transform at s_setprio(digit - 1); 0 = off (WgNtt::prio)
#endif
#ifndef HX_IINV_PRIO
#define HX_IINV_PRIO 0   // ... of an integer inverse transform
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hx {

typedef uint64_t u64;
typedef uint32_t u32;

__device__ __forceinline__ u64 mulhi(u64 a, u64 b) { return __umul64hi(a, b); }

// Harvey lazy product  W*x - hi64(W'*x)*q   (mod 2^64)   -- fwd_ntt.cpp:336-354, mod_ops.hpp:153-162
__device__ __forceinline__ u64 lazy_mul(u64 x, u64 w, u64 wp, u64 q) { return w * x - mulhi(x, wp) * q; }

__device__ __forceinline__ u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }

// The same two operations written for the gfx950 VALU (identical results mod 2^64):
//  * the subtraction W*x - Qh*q becomes an addition with nq = 2^64 - q, so both 64x64 low products share one
//    v_mad_u64_u32 accumulator chain and the four cross terms only touch the high dword (32-bit mul + add3);
//  * x - m is x + (2^64 - m): one v_lshl_add_u64 instead of a v_sub_co/v_subb_co pair with its VCC hazard nops.
// ~27 instead of ~36 VALU instructions per butterfly.
struct ModConst { u64 q, twoq, nq, n2q; };           // q, 2q, -q, -2q (mod 2^64)
__device__ __forceinline__ ModConst mod_const(u64 q) { return ModConst{q, q << 1, 0 - q, 0 - (q << 1)}; }

__device__ __forceinline__ u64 lazy_mul_n(u64 x, u64 w, u64 wp, u64 nq) {
    const u64 qh = mulhi(x, wp);
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u32 h0 = (u32)qh, h1 = (u32)(qh >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    u64 acc = (u64)w0 * x0;
    acc += (u64)h0 * n0;
    const u32 hi = (u32)(acc >> 32) + w0 * x1 + w1 * x0 + h0 * n1 + h1 * n0;
    return ((u64)hi << 32) | (u32)acc;
}
__device__ __forceinline__ u64 csub_n(u64 x, u64 m, u64 negm) {
    const u64 d = x + negm;
    return x >= m ? d : x;
}

// Twiddle loads do not depend on the loop a transform sits in (polynomial / decomposition index), so
// LICM hoists ALL of them out of that loop and the register allocator spills hundreds of VGPRs
// (measured: 0 -> 250 spills). An offset the compiler cannot see through, added inside the loop body, pins the
// loads. (Laundering the POINTER instead loses its address space: every load through it becomes a FLAT load.)
__device__ __forceinline__ u32 opaque_zero() {
    u32 z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

// Rows of 64-bit words (key rows, the next round's input) read with BUFFER loads: resource descriptor in SGPRs, the
// thread's byte offset in one VGPR, the row offset in an SGPR -- no 64-bit VALU address arithmetic (global loads at row
// strides beyond the 13-bit immediate cost a v_add_co / v_addc pair and a hazard nop each: ~6 % of the slot-major
// keyswitch kernel's VALU instructions).
template <class V>
struct RowStream {
    static_assert(sizeof(V) == 8, "64-bit words");
    __amdgpu_buffer_rsrc_t rsrc;
    // (the base is wave-uniform by construction; saying so keeps the compiler from wrapping every load in a
    // "waterfall" loop over the distinct descriptors of a wave)
    __device__ __forceinline__ static V* uniform(const V* p) {
        const unsigned long long v = (unsigned long long)p;
        const u32 lo = __builtin_amdgcn_readfirstlane(u32(v)), hi = __builtin_amdgcn_readfirstlane(u32(v >> 32));
        return (V*)(((unsigned long long)hi << 32) | lo);
    }
    __device__ __forceinline__ RowStream(const V* base, u32 bytes)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(uniform(base), 0, (int)bytes, 0x00020000)) {}
    // AUX = cache policy bits of the instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1); 2 marks a stream nobody re-reads
    template <int AUX = 0>
    __device__ __forceinline__ V at(u32 thread_byte_offset, u32 row_byte_offset) const {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)thread_byte_offset, (int)row_byte_offset, AUX);
        V d;
        __builtin_memcpy(&d, &x, 8);
        return d;
    }
};

template <int LOGN, int LOGE>
struct Geom {
    static constexpr int N = 1 << LOGN;
    static constexpr int E = 1 << LOGE;
    static constexpr int T = N / E;                        // threads per workgroup
    static constexpr int P = (LOGN + LOGE - 1) / LOGE;     // register passes
    static constexpr int KL = LOGN - (P - 1) * LOGE;       // stages in the partial pass
    static constexpr int NG = 1 << (LOGE - KL);            // independent groups in that pass
    // LDS padding: +1 word per 16 keeps 17-word lane strides conflict-free for ds_*_b64,
    // +16 words per 512 puts consecutive 512-word runs on opposite bank halves.
    static constexpr int LDS_WORDS = N + (N >> 4) + ((N >> 9) << 4);
    static constexpr size_t LDS_BYTES = size_t(LDS_WORDS) * 8;
    // half-size exchange (redeal_half): N/2 words, same padding rule. N = 32768 does not fit the CU's 160 KiB of
    // LDS in one piece and always uses it (HALF_ONLY).
    static constexpr int LDS_HALF_WORDS = (N >> 1) + (N >> 5) + ((N >> 10) << 4);
    static constexpr size_t LDS_HALF_BYTES = size_t(LDS_HALF_WORDS) * 8;
    static constexpr bool HALF_ONLY = LDS_BYTES > 160 * 1024;
    static constexpr size_t LDS_USED = HALF_ONLY ? LDS_HALF_BYTES : LDS_BYTES;

    static constexpr int LOGT = LOGN - LOGE;
    static constexpr int WB = LOGT < 6 ? LOGT : 6;         // thread-index bits that are lane bits

    __host__ __device__ static constexpr int pad(int idx) { return idx + (idx >> 4) + ((idx >> 9) << 4); }
    __host__ __device__ static constexpr int idxA(int r, int tid) { return r * T + tid; }
    // partial pass ("B order"): a thread owns NG groups of 2^KL adjacent coefficients. Lanes sit on index bits
    // [KL, KL+6), the group number above them and the wave number on the top bits -- the same top bits a wave
    // owns in every full pass with LO <= 6, so the re-deals between those passes never leave the wave.
    __host__ __device__ static constexpr int grpB(int grp, int tid) {         // coefficient index >> KL
        return ((tid >> WB) << (LOGE - KL + WB)) + (grp << WB) + (tid & ((1 << WB) - 1));
    }
    __host__ __device__ static constexpr int idxB(int r, int tid) {
        return (grpB(r >> KL, tid) << KL) + (r & ((1 << KL) - 1));
    }
    // full pass whose LOGE active index bits start at bit LO
    template <int LO>
    __host__ __device__ static constexpr int idxF(int r, int tid) {
        return ((tid >> LO) << (LO + LOGE)) + (r << LO) + (tid & ((1 << LO) - 1));
    }
    // a re-deal between two ownership maps whose upper one is idxF<LO> moves data only inside a wave when the
    // wave-number bits of the thread index map to the same coefficient bits on both sides
    template <int LO>
    static constexpr bool wave_private = (LOGT <= 6) || (LO <= 6);
};

// ---------------------------------------------------------------------------------------------
// register butterfly networks. `v` is the thread's E words, OFF the first register of the group.
// ---------------------------------------------------------------------------------------------

// forward: K stages, first one is global stage S0 (1-based, m = 2^(S0-1)); G = index bits above
// the active field. Twiddle of stage u, sub-block j:  roots[2^(S0-1+u) + (G<<u) + j].
template <int E, int OFF, int K, int S0>
__device__ __forceinline__ void fwd_stages(u64 (&v)[E], u32 G, const u64* __restrict__ roots,
                                           const u64* __restrict__ precon, u64 q, u64 twoq) {
    const ModConst mc = mod_const(q);
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const u32 base = (1u << (S0 - 1 + u)) + (G << u);
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const u64 W = roots[base + j];
            const u64 Wp = precon[base + j];
#pragma unroll
            for (int c = 0; c < (1 << (K - 1 - u)); ++c) {
                const int a0 = OFF + (j << (K - u)) + c;
                const int a1 = a0 + (1 << (K - 1 - u));
                const u64 X = v[a0], Y = v[a1];
                const u64 tx = csub_n(X, twoq, mc.n2q);    // fwd_ntt.cpp:322-323
                const u64 Q = lazy_mul_n(Y, W, Wp, mc.nq); // :336-354
                v[a0] = tx + Q;                            // :359
                v[a1] = tx + twoq - Q;                     // :360
            }
        }
    }
}

// inverse: K stages on index bits [LO, LO+K); stage u has t = 2^(LO+u);
// twiddle index  N - N/2^(LO+u) + 1 + (G << (K-1-u)) + j   (root_index walk of inv_ntt.cpp:141-306).
// If LAST, the final stage (t = N/2) is the fused n^-1 scaling of inv_ntt.cpp:400-437.
template <int E, int OFF, int K, int LO, int LOGN, bool LAST>
__device__ __forceinline__ void inv_stages(u64 (&v)[E], u32 G, const u64* __restrict__ iroots,
                                           const u64* __restrict__ iprecon, u64 q, u64 twoq,
                                           u64 inv_n, u64 inv_n_p, u64 inv_n_w, u64 inv_n_w_p) {
    constexpr u32 N = 1u << LOGN;
    const ModConst mc = mod_const(q);
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const bool fused = LAST && (u == K - 1);
        const u32 base = N - (N >> (LO + u)) + 1 + (G << (K - 1 - u));
#pragma unroll
        for (int j = 0; j < (1 << (K - 1 - u)); ++j) {
            u64 W = 0, Wp = 0;
            if (!fused) { W = iroots[base + j]; Wp = iprecon[base + j]; }
#pragma unroll
            for (int c = 0; c < (1 << u); ++c) {
                const int a0 = OFF + (j << (u + 1)) + c;
                const int a1 = a0 + (1 << u);
                const u64 X = v[a0], Y = v[a1];
                const u64 tx = csub_n(X + Y, twoq, mc.n2q);  // inv_ntt.cpp:300-304
                const u64 ty = X + twoq - Y;
                if (!fused) {
                    v[a0] = tx;
                    v[a1] = lazy_mul(ty, W, Wp, q);          // :305-306 (the fused-accumulator form compiles worse here)
                } else {
                    v[a0] = csub_n(lazy_mul_n(tx, inv_n, inv_n_p, mc.nq), q, mc.nq);      // :413-432
                    v[a1] = csub_n(lazy_mul_n(ty, inv_n_w, inv_n_w_p, mc.nq), q, mc.nq);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS re-deal: write registers under the current ownership map, read under the next one.
// ---------------------------------------------------------------------------------------------
template <class G, class FromIdx, class ToIdx>
__device__ __forceinline__ void redeal(u64 (&v)[G::E], u64* lds, int tid, FromIdx from, ToIdx to) {
#pragma unroll
    for (int r = 0; r < G::E; ++r) lds[G::pad(from(r, tid))] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = lds[G::pad(to(r, tid))];
    __syncthreads();
}

// compiler-only ordering point for LDS traffic that stays inside one wave (the hardware executes a wave's LDS
// instructions in order; this keeps the compiler from moving reads above the writes of other lanes)
__device__ __forceinline__ void wave_fence() { asm volatile("" ::: "memory"); }

// Re-deal with the minimum of synchronisation. PRIVATE: every coefficient stays inside its wave -> no
// s_barrier at all, the waves of the workgroup drift apart and cover each other's LDS/memory latency.
// Otherwise one barrier between writes and reads, preceded (LEAD) by one that waits for earlier readers of
// the words about to be overwritten. No trailing barrier: after a cross-wave exchange a wave only ever
// touches its own block until the next cross-wave exchange, which brings its own LEAD barrier.
// Every ownership map is "thread part OR register part" on disjoint coefficient bits, and Geom::pad only shifts
// and adds, so pad(at(r, tid)) = pad(at(0, tid)) + pad(at(r, 0)): one address per side plus compile-time offsets
// (the LDS instructions' immediate field) instead of ~7 integer instructions per coefficient.
// (Un-pairing the accesses -- MI355X_MICROARCH.md's LDS table has ds_read2_b64 at 8 cycles per pair against 2 + 2 -- measured -0.6 % with
// hand-written ds_read_b64 and far worse through volatile accesses: tools/experiments/README.md, round 4.)
template <class G, bool PRIVATE, bool LEAD, class V, class FromIdx, class ToIdx>
__device__ __forceinline__ void redeal_x(V (&v)[G::E], V* lds, int tid, FromIdx from, ToIdx to) {
    V* const wr = lds + G::pad(from(0, tid));
    V* const rd = lds + G::pad(to(0, tid));
    if constexpr (PRIVATE) wave_fence();
    else if constexpr (LEAD) __syncthreads();
#pragma unroll
    for (int r = 0; r < G::E; ++r) wr[G::pad(from(r, 0))] = v[r];
    if constexpr (PRIVATE) wave_fence();
    else __syncthreads();
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = rd[G::pad(to(r, 0))];
    if constexpr (PRIVATE) wave_fence();
}

// Half-size re-deal: the exchange goes through LDS in two rounds, so a transform needs N/2 words of LDS (N = 32768
// in 140 KiB). Works between two ownership maps that swap the top register bit with thread bit SEL: coefficient bits
// (a, b) with a <-> register top bit in `from` and thread bit SEL in `to`, b the other way round. Round h moves the
// coefficients with bit a XOR bit b == h; for a thread whose bit SEL is t that is register half (h XOR t) under BOTH
// maps, so every round reads back into the registers it has just sent. Threads with t = 1 swap their register halves
// before and after, so the LDS traffic itself is the same straight-line code for every lane (only the base addresses
// depend on t). DROP = max(a, b) is the bit left out of the LDS slot number. SEL >= 6: whole waves take the same
// side, barriers between the phases (LEAD as in redeal_x; TRAIL because a wave's private block overlaps words other
// waves read in round 1). SEL < 6: the exchange stays inside the wave, no barriers.
// (Measured for N = 16384 as a way to fit two workgroups per CU: bit-exact, no faster -- tools/experiments/.)
template <class G, int DROP>
__device__ __forceinline__ int half_slot(int idx) {
    return G::pad(((idx >> (DROP + 1)) << DROP) | (idx & ((1 << DROP) - 1)));
}
template <class G, class V>
__device__ __forceinline__ void swap_halves_if(V (&v)[G::E], bool t) {
#pragma unroll
    for (int r = 0; r < G::E / 2; ++r) {
        const V lo = v[r], hi = v[r + G::E / 2];
        v[r] = t ? hi : lo;
        v[r + G::E / 2] = t ? lo : hi;
    }
}
template <class G, int SEL, int DROP, bool LEAD, bool TRAIL, class V, class FromIdx, class ToIdx>
__device__ __forceinline__ void redeal_half(V (&v)[G::E], V* lds, int tid, FromIdx from, ToIdx to) {
    constexpr bool PRIVATE = SEL < 6;
    constexpr int H = G::E / 2;
    const bool t = (tid >> SEL) & 1;
    // slot distance between a register and its partner in the other half (the partner bit is clear in at(r < H))
    const int dfrom = half_slot<G, DROP>(from(H, tid)) - half_slot<G, DROP>(from(0, tid));
    const int dto = half_slot<G, DROP>(to(H, tid)) - half_slot<G, DROP>(to(0, tid));
    // physical register r < H holds logical register r + H*t, physical r + H holds logical r + H*(1 - t)
    V* const wr0 = lds + (t ? dfrom : 0);
    V* const wr1 = lds + (t ? 0 : dfrom);
    V* const rd0 = lds + (t ? dto : 0);
    V* const rd1 = lds + (t ? 0 : dto);
    swap_halves_if<G>(v, t);
    if constexpr (PRIVATE) wave_fence();
    else if constexpr (LEAD) __syncthreads();
#pragma unroll
    for (int r = 0; r < H; ++r) wr0[half_slot<G, DROP>(from(r, tid))] = v[r];
    if constexpr (PRIVATE) wave_fence(); else __syncthreads();
#pragma unroll
    for (int r = 0; r < H; ++r) v[r] = rd0[half_slot<G, DROP>(to(r, tid))];
    if constexpr (PRIVATE) wave_fence(); else __syncthreads();
#pragma unroll
    for (int r = 0; r < H; ++r) wr1[half_slot<G, DROP>(from(r, tid))] = v[r + H];
    if constexpr (PRIVATE) wave_fence(); else __syncthreads();
#pragma unroll
    for (int r = 0; r < H; ++r) v[r + H] = rd1[half_slot<G, DROP>(to(r, tid))];
    if constexpr (PRIVATE) wave_fence();
    else if constexpr (TRAIL) __syncthreads();
    swap_halves_if<G>(v, t);
}

// Readers' gate (round 6). An INVERSE transform ends with its cross-wave exchange: behind the barrier every wave reads words out of every
// other wave's block. The transform that follows is protected if it is a forward one (its first LDS access is a cross-wave exchange with a
// LEAD barrier) -- but another INVERSE transform starts with a wave-PRIVATE re-deal, whose writes go to the wave's own block without any
// barrier: a wave that runs ahead overwrites words a slower wave of the workgroup has not read yet. With one 16-wave workgroup per CU the
// waves of a SIMD share a phase and a priority and the window (a pass, a store and a partial pass: > 500 instructions) never closed; with
// several small workgroups per CU it does -- a wave at the low per-pass priority starves behind other workgroups' waves: N = 2048,
// s'_0 of one instance in a few thousand wrong (tools/soak_ks_random.py found it). A barrier in front of every inverse transform would put
// the waves of the persistent kernels back in lockstep once per polynomial, so the two halves of one are split instead: a wave ARRIVES
// (one LDS atomic by lane 0) right behind its cross-wave reads -- a wave's LDS instructions execute in order, so they have been served when
// the increment is -- and WAITS in front of its next inverse transform's first private write until every wave of the workgroup has arrived
// as often as it has itself. Arrivals precede the wait by hundreds of instructions: the wait all but never spins, and no wave can wait for
// an arrival that is behind a barrier it has not passed itself (no deadlock). The counter is the last word of the exchange array, which
// Geom::pad never reaches (pad(N - 1) = LDS_WORDS - 18).
struct NoGate {
    __device__ __forceinline__ void arrive() const {}
    __device__ __forceinline__ void wait() const {}
};
template <class G>
struct ReadersGate {
    static_assert(G::pad(G::N - 1) < G::LDS_WORDS - 1, "the counter's word must be outside the exchange slots");
    static constexpr u32 WAVES = (G::T + 63) / 64;
    // (half-size exchanges -- N = 32768 in one workgroup -- are not gated: their slots change from re-deal to re-deal; the one kernel that runs
    // inverse after inverse on them, k_ntt_inv_p<15, 5> under HEXL_NTT_HALVES=0, has a barrier per polynomial. One-wave workgroups need nothing.)
    static constexpr bool ON = !G::HALF_ONLY && WAVES > 1;
    u32* cnt;
    u32 expect;                                                   // arrivals every wave must have made before this one overwrites its block
    // (one barrier per kernel; `lds` = the exchange array)
    __device__ __forceinline__ explicit ReadersGate(void* lds) : cnt(reinterpret_cast<u32*>(static_cast<u64*>(lds) + (G::LDS_WORDS - 1))), expect(0) {
        if constexpr (ON) {
            if (threadIdx.x == 0) *cnt = 0;
            __syncthreads();
        }
    }
    __device__ __forceinline__ void arrive() {
        if constexpr (ON) {
            asm volatile("" ::: "memory");
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            expect += WAVES;
        }
    }
    __device__ __forceinline__ void wait() {
        if constexpr (ON) {
            asm volatile("" ::: "memory");
            while (int(u32(__builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) - expect) < 0)
                __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    }
};
// where a transform arrives / waits: arrive behind a cross-wave exchange of an inverse transform, wait in front of its first re-deal
template <class Gate> inline constexpr bool gate_on = true;
template <> inline constexpr bool gate_on<NoGate> = false;
// An inverse transform that is not the only LDS user of its kernel (FRESH = false) has to NAME what orders it behind an earlier inverse
// transform's cross-wave reads -- a ReadersGate, or this tag: "a forward transform or a barrier of the caller lies in between".
// (inverse<false> with neither does not compile: the hole of round 6 was a default.)
struct OrderedByCaller : NoGate {};
template <> inline constexpr bool gate_on<OrderedByCaller> = false;
template <bool FRESH, class Gate> inline constexpr bool inverse_is_ordered = FRESH || gate_on<Gate> || __is_same(Gate, OrderedByCaller);

// The re-deal between the full pass on coefficient bits [LO, LO+LOGE) (`idxF<LO>`) and its lower neighbour -- the
// full pass below it, or the partial pass (B order) when LOWER_IS_B -- in the direction of the transform.
template <class G, int LO, int LOGE, bool FORWARD, bool LEAD, bool LOWER_IS_B, class V>
__device__ __forceinline__ void redeal_pass(V (&v)[G::E], V* lds, int tid) {
    auto upper = [](int r, int t) { return G::template idxF<LO>(r, t); };
    auto lower = [](int r, int t) {
        if constexpr (LOWER_IS_B) return G::idxB(r, t);
        else return G::template idxF<LO - LOGE>(r, t);
    };
    if constexpr (G::HALF_ONLY) {
        // the lower map must carry the top register bit on coefficient bit LO-1: true for a full pass, and for B only
        // when the partial pass is a full one as well (NG == 1, B == idxF<0>)
        static_assert(!LOWER_IS_B || (G::NG == 1 && LO == LOGE), "half-size exchange: unsupported geometry");
        if constexpr (FORWARD) redeal_half<G, LO - 1, LO + LOGE - 1, LEAD, true>(v, lds, tid, upper, lower);
        else                   redeal_half<G, LO - 1, LO + LOGE - 1, true, false>(v, lds, tid, lower, upper);
    } else {
        constexpr bool PRIV = G::template wave_private<LO>;
        if constexpr (FORWARD) redeal_x<G, PRIV, LEAD>(v, lds, tid, upper, lower);
        else                   redeal_x<G, PRIV, LEAD>(v, lds, tid, lower, upper);
    }
}

template <int LOGN, int LOGE>
struct WgNtt {
    using G = Geom<LOGN, LOGE>;
    static constexpr int E = G::E;

    // experiment knobs, as HX_FWD_PRIO / HX_INV_PRIO of ntt_core_f64.hpp: wave priority by pass of the INTEGER transforms
    template <int KNOB, int PH>
    __device__ static __forceinline__ void prio() {
        if constexpr (KNOB != 0 && PH < 4) {
            constexpr int d = PH == 0 ? KNOB / 1000 : PH == 1 ? (KNOB / 100) % 10 : PH == 2 ? (KNOB / 10) % 10 : KNOB % 10;
            __builtin_amdgcn_s_setprio(d - 1);
        }
    }
    // ---- forward: v in A layout on entry, B layout on exit; values in [0,4q) (lazy) ---------
    // FRESH: no other LDS traffic of this workgroup can still be in flight (single-transform kernels)
    template <int PASS, bool FRESH = false>
    __device__ static __forceinline__ void fwd_pass(u64 (&v)[E], u64* lds, int tid,
                                                    const u64* roots, const u64* precon, u64 q, u64 twoq) {
        prio<HX_IFWD_PRIO, PASS>();
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = LOGN - (PASS + 1) * LOGE;
            // pass 0 has no index bits above its field: constant twiddle addresses -> scalar loads
            // LO >= 6: every lane of a wave shares the group index -> scalar twiddle loads
            const u32 Gp = (PASS == 0) ? 0u : (LO >= 6 ? u32(__builtin_amdgcn_readfirstlane(u32(tid) >> LO)) : (u32(tid) >> LO));
            fwd_stages<E, 0, LOGE, PASS * LOGE + 1>(v, Gp, roots, precon, q, twoq);
            // hand over to the next pass's ownership
            constexpr bool LEAD = !(FRESH && PASS == 0);
            redeal_pass<G, LO, LOGE, true, LEAD, (PASS + 1 == G::P - 1)>(v, lds, tid);
            fwd_pass<PASS + 1, FRESH>(v, lds, tid, roots, precon, q, twoq);
        } else {
            // partial pass: NG groups of 2^KL contiguous words
            fwd_last<0>(v, tid, roots, precon, q, twoq);
        }
    }
    template <int GRP>
    __device__ static __forceinline__ void fwd_last(u64 (&v)[E], int tid, const u64* roots,
                                                    const u64* precon, u64 q, u64 twoq) {
        if constexpr (GRP < G::NG) {
            const u32 Gbits = u32(G::grpB(GRP, tid));
            fwd_stages<E, GRP * (1 << G::KL), G::KL, (G::P - 1) * LOGE + 1>(v, Gbits, roots, precon, q, twoq);
            fwd_last<GRP + 1>(v, tid, roots, precon, q, twoq);
        }
    }
    template <bool FRESH = false>
    __device__ static __forceinline__ void forward_lazy(u64 (&v)[E], u64* lds, int tid, const u64* roots,
                                                        const u64* precon, u64 q) {
        fwd_pass<0, FRESH>(v, lds, tid, roots, precon, q, q << 1);
    }
    // fwd_ntt.cpp:369-384
    __device__ static __forceinline__ void final_reduce(u64 (&v)[E], u64 q) {
        const ModConst mc = mod_const(q);
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = csub_n(csub_n(v[r], mc.twoq, mc.n2q), q, mc.nq);
    }

    // ---- inverse: v in B layout on entry, A layout on exit; values in [0,q) -----------------
    template <int GRP>
    __device__ static __forceinline__ void inv_first(u64 (&v)[E], int tid, const u64* iroots,
                                                     const u64* iprecon, u64 q, u64 twoq, u64 a, u64 ap,
                                                     u64 b, u64 bp) {
        if constexpr (GRP < G::NG) {
            const u32 Gbits = u32(G::grpB(GRP, tid));
            inv_stages<E, GRP * (1 << G::KL), G::KL, 0, LOGN, (G::P == 1)>(v, Gbits, iroots, iprecon, q,
                                                                          twoq, a, ap, b, bp);
            inv_first<GRP + 1>(v, tid, iroots, iprecon, q, twoq, a, ap, b, bp);
        }
    }
    // PASS counts the full passes after the partial one: active bits [KL + PASS*LOGE, +LOGE)
    // `gate` (ReadersGate above): for an inverse transform that may FOLLOW another one in the same workgroup, and for the one it follows
    template <int PASS, bool FRESH = false, class Gate = NoGate>
    __device__ static __forceinline__ void inv_pass(u64 (&v)[E], u64* lds, int tid, const u64* iroots,
                                                    const u64* iprecon, u64 q, u64 twoq, u64 a, u64 ap,
                                                    u64 b, u64 bp, Gate* gate = nullptr) {
        if constexpr (PASS < G::P - 1) {
            constexpr int LO = G::KL + PASS * LOGE;
            constexpr bool LEAD = !(FRESH && PASS == 0);
            prio<HX_IINV_PRIO, PASS + 1>();
            if constexpr (gate_on<Gate> && PASS == 0 && !G::HALF_ONLY) gate->wait();
            redeal_pass<G, LO, LOGE, false, LEAD, (PASS == 0)>(v, lds, tid);
            if constexpr (gate_on<Gate> && PASS == G::P - 2 && !G::HALF_ONLY && !G::template wave_private<LO>) gate->arrive();
            const u32 Gp = (PASS == G::P - 2) ? 0u : (LO >= 6 ? u32(__builtin_amdgcn_readfirstlane(u32(tid) >> LO)) : (u32(tid) >> LO));
            inv_stages<E, 0, LOGE, LO, LOGN, (PASS == G::P - 2)>(v, Gp, iroots, iprecon, q, twoq, a, ap, b, bp);
            inv_pass<PASS + 1, FRESH, Gate>(v, lds, tid, iroots, iprecon, q, twoq, a, ap, b, bp, gate);
        }
    }
    template <bool FRESH = false, class Gate = NoGate>
    __device__ static __forceinline__ void inverse(u64 (&v)[E], u64* lds, int tid, const u64* iroots,
                                                   const u64* iprecon, u64 q, u64 inv_n, u64 inv_n_p,
                                                   u64 inv_n_w, u64 inv_n_w_p, Gate* gate = nullptr) {
        static_assert(inverse_is_ordered<FRESH, Gate>, "inverse<false>: pass a ReadersGate, or OrderedByCaller if a forward transform or a barrier precedes");
        const u64 twoq = q << 1;
        prio<HX_IINV_PRIO, 0>();
        inv_first<0>(v, tid, iroots, iprecon, q, twoq, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
        inv_pass<0, FRESH, Gate>(v, lds, tid, iroots, iprecon, q, twoq, inv_n, inv_n_p, inv_n_w, inv_n_w_p, gate);
    }
};

// floor(y * 2^64 / q) computed on the device once per launch (the reference's
// MultiplyUIntModLazy3 divides per element, mod_ops.hpp:143-145): binary long division, y < q.
__device__ __forceinline__ u64 shoup_factor(u64 y, u64 q) {
    u64 rem = y, quo = 0;
    for (int i = 0; i < 64; ++i) {
        const bool carry = rem >> 63;
        rem <<= 1;
        quo <<= 1;
        if (carry || rem >= q) { rem -= q; quo |= 1; }
    }
    return quo;
}

}  // namespace hx
