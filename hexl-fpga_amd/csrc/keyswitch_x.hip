// keyswitch_x.hip -- K4 on the FP64 pipe, SLOT-MAJOR pipeline for large batches (N = 1024 .. 16384): the key
// multiply-accumulate (device/keyswitch/dyadmult.hpp:85-166) runs in the registers of the
// workgroup that produced the transforms, as the reference's pipes do -- `u` (15 MB per keyswitch in the (b, d)-major
// pipeline of keyswitch_f64.hip) and `prod` never exist in memory.
//
// One workgroup = N/16 threads x 16 coefficients (Geom<LOGN,4>: 1024 threads at N = 16384; 128 VGPRs per thread): 32
// registers hold the polynomial in flight, 64 the two accumulators (prod[k][slot], k = 0, 1), the rest is working space.
// Three kernels per chunk:
//
//   k_ksx_intt    (b, d)    c_d = INTT_{q_d}(t_target[d])  -> scratch (canonical doubles, natural order)       (step 1)
//   k_ksx_special (b)       acc_k = sum_d NTT_{q_sp}(c_d mod q_sp) . key[d][special][k]                       (steps 2-3)
//                           s'_k = INTT_{q_sp}(acc_k) + floor(q_sp/2)  -> scratch                              (step 4)
//   k_ksx_main    (b, i<L)  acc_k = sum_d NTT_{q_i}(c_d mod q_i) . key[d][i][k]   (d == i: t_target[i] itself)
//                           k = 0, 1:  w = NTT_{q_i}((s'_k + fix_i) mod q_i);  result[k][i] += (acc_k - w) . msf_i  (steps 5-7)
//
// HBM-side traffic per keyswitch (PMC, profiles/): 14.5 MB against 25.6 MB for the (b, d)-major pipeline and 4.6 MB
// algorithmic. Arithmetic and bounds are those of f64_arith.hpp; results are bit-identical to the (b, d)-major
// pipeline and the integer kernels.
//
// What shaped the code (all measured on the MI355X, tools/ksx_timeline.hip; numbers in DESIGN.md 4.4):
//  * REGISTERS. Everything lives or dies by keeping the 64 accumulator registers out of scratch: a spilled accumulator
//    is reloaded inside the multiply-accumulate, and because vector memory returns in order that reload waits for the
//    whole key prefetch queue -- 37-45 k cycles per multiply-accumulate instead of 10 k. Hence: straight-line phases
//    instead of one loop with an up/down branch (128 phi nodes on the accumulators cost ~200 spills), the inverse
//    transforms in their own kernel (two twiddle tables: 92 working registers), no persistent item loop around k_ksx_main,
//    a key ring of three pairs, not six.
//  * 16 x 1024 beats 32 x 512 (256 VGPRs, two re-deals instead of three; 173 k against 209 k keyswitch/s on the round-4 kernels): with
//    two waves per SIMD every exposed load latency is paid in full. The 32 x 512 variants are gone from the build (round 6).
//  * The NEXT round's input is requested inside the multiply-accumulate, into the registers the products free.
//  * Per-lane twiddles are requested by hand ahead of their butterflies (KX_PRE, KX_IPRE): with 96 data registers the
//    compiler otherwise puts each load in front of its first use and waits for it on the spot.
//  * Round 4, instructions that were not needed (the pipeline is bound by FP64 issue): when the moduli of a
//    plan are within a factor LAZY_SKIP_MAX_RATIO of each other (SKIP kernels; always true for the reference's parameter sets
//    of equal-sized primes) c_d enters the mod-up transforms as it is -- canonical below q_d, i.e. at most 1.25 q_i -- on a
//    reduction schedule shifted by one stage (f64_arith.hpp), instead of being range-reduced first; s' is stored as the
//    exact centred remainder y = s'_canonical - floor(q_sp/2) (what intt2_redu.hpp's "+ fix" turns it into anyway), so the
//    mod-down transforms take it as it is too; the d == i term and the accumulators go un-reduced into mac_fold / the mod-down
//    epilogue (bounds there); 64-bit words become doubles by an OR and a subtraction; the inverse transforms take their
//    quotients from the products (no w/p table: half the per-lane twiddle bytes). 5.7 % fewer VALU instructions per keyswitch.
#include <stdlib.h>

// Wave priority by pass of the forward transforms (ntt_core_f64.hpp hx_fwd_prio; round 4): the pass in front of the cross-wave
// barrier runs at priority 0, everything behind it at 1. The four waves of a SIMD are staggered by up to a barrier interval (the
// oldest wave wins every issue slot, tools/ksx_timeline gantt): a wave that is a round ahead only reaches the barrier to wait there,
// while the wave it waits for is still finishing the previous round's passes and multiply-accumulate on the same SIMD -- the leader's
// first pass should fill the laggard's stalls, not compete with it. +2.4 ... 2.9 % on three boxes (tools/experiments/README.md).
#ifndef HX_FWD_PRIO
#define HX_FWD_PRIO 1222
#endif
#include "hexl_internal.hpp"
#include "ntt_core_f64.hpp"

using namespace hx;

// Tuning constants that were swept on the MI355X and may want re-sweeping on another ROCm (tools/build_variant.sh <name> -D...); every
// other knob of rounds 2-5 is gone from the source: its experiment is in tools/experiments/ (README.md + r06_pruned_knobs_*.patch).
#ifndef KX_PRE
#define KX_PRE 11     // forward transforms: twiddles of the per-lane passes requested early (ntt_core_f64.hpp WgNttF64 PRE): units = groups of
                      // the last pass ahead of its re-deal, tens = early stages of the per-lane full pass up front
#endif
#ifndef KX_IPRE
#define KX_IPRE 1     // k_ksx_intt: the per-lane twiddle pairs of a pass of the inverse transform requested before its butterflies
#endif

// 16 coefficients per thread: the kernels are written for 128 VGPRs = four waves per SIMD, which a 1024-thread workgroup
// (N = 16384) implies and the smaller ring dimensions (512 ... 64 threads, several workgroups per CU) have to ask for
#define KX_WAVES(LOGE) ((LOGE) == 4 ? 4 : 2)

// Profiling aids (WRONG RESULTS; -DHEXL_PROFILING_AIDS builds only: lib/libhexl_mi355x_prof.so, never the shipped library):
// HEXL_KSX_ALIAS=<bit mask> makes every workgroup read one instance's / limb's rows of a stream, which takes that stream out
// of the L2-miss-side counters and of the fabric's power draw (profiles/r04_bytes.json): 1 = key rows (every row reads row 0),
// 2 = c and s' reads (instance 0), 4 = t_target reads (instance 0), 8 = result read-modify-write (instance 0), 16 = twiddle
// tables (limb 0).
#ifdef HEXL_PROFILING_AIDS
#define KX_ALIASED(bit, x) ((a.alias & (bit)) ? 0u : (x))
#else
#define KX_ALIASED(bit, x) (x)
#endif

struct KsArgsX {
    const KsModF64* mods;    // [K]
    const double* tables;    // [K][4][n]: w, w/p, inverse w (first entry at index 1), inverse w/p
    const double* keys;      // [L][L+1][2][n] centred, B order of THIS geometry
    double* c;               // [chunk][L][n]   canonical, natural order
    double* s;               // [chunk][2][n]   canonical, natural order
    // N = 32768 (k_ksh_* kernels): the two 16384-point sub-inverses of every inverse transform, un-scaled, natural order inside each half
    double* csub;            // [chunk][L][2][n/2]
    double* ssub;            // [chunk][2][2][n/2]
    const u64* t_target;     // [chunk][L][n]
    u64* result;             // [chunk][2][L][n]
    u32 L, K, nb;
    // the limbs THIS launch works on (k_ksx_intt: d, k_ksx_main: i): limb number j < nsel is nibble j of selmap. A plan whose moduli
    // share one arithmetic tier launches once with nsel = L and the identity map; a plan of mixed tiers (hexl_ks_plan::mixed) launches
    // each kernel once per tier with that tier's limbs -- a workgroup transforms modulo ONE q_i, so its reduction period is that limb's
    u32 nsel;
    unsigned long long selmap;
    u32* range_flag;         // set to 1 when a t_target / result word is not below its modulus (hexl_ks_range_check)
    u32 key_stride;          // words between key[d][slot] rows: 2 n (profiling builds, alias bit 1: 0)
    u32 alias;               // profiling builds only (KX_ALIASED); 0 in the shipped library
    // fused multiply + relinearize (hexl_multiply_relinearize): ciphertext pairs a, b [chunk][2][L][n]; the keyswitch input
    // is a_1 . b_1 (never stored) and `result` is WRITTEN with (a_0 b_0, a_0 b_1 + a_1 b_0) + keyswitch(a_1 b_1)
    const u64 *mul_a, *mul_b;
    unsigned long long* stamps;   // tools/ksx_timeline.hip only (KX_TIMELINE builds): [workgroup][wave][KX_NST]
};

// Optional per-wave cycle stamps at the phase boundaries of a round (tools/ksx_timeline.hip). No waits are forced:
// a stamp shows when the wave's instruction stream got there.
#ifdef KX_TIMELINE
constexpr int KX_NST = 64;
#define KX_STAMP(i)                                                                                            \
    do {                                                                                                       \
        _Pragma("unroll") for (int r_ = 0; r_ < G::E; ++r_) asm volatile("" : "+v"(v[r_]));                    \
        if (a.stamps && (threadIdx.x & 63) == 0)                                                               \
            a.stamps[(size_t(kx_slot) * (G::T / 64) + (threadIdx.x >> 6)) * KX_NST + (i)] = __builtin_readcyclecounter(); \
        _Pragma("unroll") for (int r_ = 0; r_ < G::E; ++r_) asm volatile("" : "+v"(v[r_]));                    \
    } while (0)
#else
#define KX_STAMP(i) do { } while (0)
#endif

// Persistent workgroups: the grid is 8 x g workgroups (g per XCD, normally one per CU); the g workgroups of XCD x
// (block index & 7, MI355X_MICROARCH.md) walk that XCD's contiguous share of the item list side by side, so that items
// which read the same intermediate run at the same time on the same L2. A workgroup that stays on its CU does not pay
// the dispatch gap between workgroups (LDS and all 512 registers per SIMD lane are only handed over when the LAST wave
// of the previous workgroup has finished: ~20 k cycles of a 360 k-cycle item), and its early waves already request
// the next item's first input.
struct XcdWalk { u32 pos, end, step; };
__device__ __forceinline__ XcdWalk xcd_walk(u32 total) {
    const u32 g = gridDim.x >> 3, q = total >> 3, r = total & 7, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const u32 start = x * q + (x < r ? x : r);
    return XcdWalk{start + j, start + q + (x < r ? 1u : 0u), g};
}

__device__ __forceinline__ u32 sel_limb(const KsArgsX& a, u32 j) { return u32(a.selmap >> (4 * j)) & 15u; }

// the per-modulus constants of limb i, read through the constant address space (ten doubles: scalar loads)
__device__ __forceinline__ KsModF64 load_mod_const(const KsModF64* p) {
    static_assert(sizeof(KsModF64) == 10 * sizeof(double), "KsModF64 is ten doubles");
    const ctw_t q = (ctw_t)(const double*)p;
    KsModF64 f;
    f.m.p = q[0]; f.m.pinv = q[1];
    f.sc.n = q[2]; f.sc.n_p = q[3]; f.sc.nw = q[4]; f.sc.nw_p = q[5];
    f.msf = q[6]; f.msf_p = q[7]; f.fix = q[8]; f.half = q[9];
    return f;
}

// key[d][slot][0] (key[d][slot][1] follows it, n words further)
template <class G>
__device__ __forceinline__ const double* key_row(const KsArgsX& a, u32 d, u32 slot) {
    return a.keys + size_t(d * (a.L + 1) + slot) * a.key_stride;
}

__device__ __forceinline__ u32 xcd_item_x(u32 bid, u32 total) {   // XCD-contiguous work ranges (keyswitch.hip)
    const u32 q = total >> 3, r = total & 7, xcd = bid & 7, j = bid >> 3;
    return xcd * q + (xcd < r ? xcd : r) + j;
}

// natural-order words -> centred doubles in B register order. With at most four adjacent words per lane (16
// coefficients per thread) the loads go straight to the B positions, as in k_ksf_up; with sixteen (32 per thread) a
// wave instruction would touch 64 cache lines, so the loads are A order (a wave reads 512 contiguous bytes per
// register) and the A -> B exchange is a cross-wave re-deal through LDS.
template <class G>
__device__ __forceinline__ void load_natural_to_B(double (&v)[G::E], const u64* __restrict__ src, double* lds, int tid,
                                                  const Mod m, hxf::RangeMask& bad) {
    if constexpr (G::KL <= 2) {
        const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(hxf::to_f64_checked((src + G::idxB(r, 0))[tB], m, bad), m);
    } else {
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(hxf::to_f64_checked((src + G::idxA(r, 0))[u32(tid)], m, bad), m);
        redeal_x<G, false, true>(v, lds, tid, [](int r, int t) { return G::idxA(r, t); },
                                 [](int r, int t) { return G::idxB(r, t); });
    }
}

// acc_k += v . key_k for the thread's E coefficients, bringing the accumulators back to |x| <= p/2 + 2 every term.
// LAZY bounds (f64_arith.hpp): |v| <= 2.14p, |key| <= p/2  =>  |v.key mod p| <= 1.31p by mul_mod, so
// |acc + product| <= 1.81p; strict kernels (moduli up to 2^52, |v| <= p/2 + 2): 0.5p + 0.7p < 2p.
// (Reducing only every second term under a wave-uniform branch costs more than it saves: the two copies of the
// accumulators meet in 128 phi nodes and the allocator spills ~200 registers.)
// The keys stream through a ring of PF register pairs requested PF coefficients ahead; as soon as v[r] is consumed
// its register receives word r of the NEXT round's input (`next`, A order; never null), so that input crosses
// the memory system during this multiply-accumulate instead of stalling the next transform. A scheduling barrier per
// coefficient keeps the compiler from hoisting the whole stream to the top (and spilling what it displaced).
// (Requesting the next input in one burst behind the last key instead measured the same within noise at four waves per
// SIMD: 10.4 k against 9.5 k cycles per multiply-accumulate in the timeline tool.)
// Ring depth: three pairs measured best (192 k keyswitch/s against 181 k for six and 165 k for eight): the other waves of
// the SIMD cover the key latency, the registers of a deeper ring are not free -- with three, k_ksx_main spills nothing INSIDE its
// round loop (compiler, round 4: 5 VGPRs / 16 bytes of scratch per lane in all, one dword reloaded at the top of every round and
// the rest between the two mod-down rounds: hipcc -Rpass-analysis=kernel-resource-usage, tools/kres.sh).
#ifndef KX_PF_DEPTH
#define KX_PF_DEPTH 3
#endif
constexpr int KX_PF = KX_PF_DEPTH;
// The two streams of a multiply-accumulate (key rows, next input rows) are read with BUFFER loads (RowStream, ntt_core.hpp).
// acc_k += v . key_k; k0 points at key[d][slot][0], key[..][1] follows it (n words further); `next` = the next round's
// input, A order (never null)
// strict kernels (moduli above the lazy bound): the semi-strict schedule in the wave-uniform passes of their forward transforms
// (ntt_core_f64.hpp SEMIU; round 5: +1.6 %). The all-passes variant (round 4) lost 12 % to its per-lane w/p loads and is gone.
#ifndef KX_SEMI_UNI
#define KX_SEMI_UNI 1
#endif
#define KX_SEMIU_ON(LAZY) ((LAZY) == 0 && KX_SEMI_UNI != 0)
// lazy kernels, N <= 16384: the forward transforms run the X schedules of f64_arith.hpp (range reduction of the added operand only, where the bound
// chain needs it) instead of the periodic full reductions: 18 / 21 instead of 24 reduction instructions per butterfly column at N = 16384 in
// the top tier (mod-up / mod-down), 2.5 % of a keyswitch's instructions. 0 = the periodic schedules (A/B: tools/build_variant.sh).
#ifndef KX_XSCHED
#define KX_XSCHED 1
#endif
// ... and the inverse transforms (k_ksx_intt, the special slot's two) the I schedules: sums and products range-reduced by the history of the
// butterfly's inputs instead of every sum at every stage (N = 16384, top tier: 31.5 instead of 45 reduction instructions per butterfly column)
#ifndef KX_ISCHED
#define KX_ISCHED 1
#endif
// FOLD: the folded multiply-accumulate (f64_arith.hpp mac_fold; lazy tiers: accumulators <= 1.6p between rounds, strict tier <= 0.9p)
// CSW: words between the two key components of a row (0: G::N; the N = 32768 kernels work on HALF rows of rows that are 2 G::N long)
template <class G, bool FOLD = false, int CSW = 0>
__device__ __forceinline__ void mac_keys(double (&acc0)[G::E], double (&acc1)[G::E], double (&v)[G::E],
                                         const double* __restrict__ k0, const double* __restrict__ next, int tid,
                                         const Mod m) {
    constexpr int PF = KX_PF;
    constexpr int CS = CSW ? CSW : G::N;
    const RowStream<double> keys(k0, 2 * CS * 8), nxt(next, G::N * 8);
    const u32 toff = u32(tid) * 8;
    double ka[PF], kb[PF];
#pragma unroll
    for (int r = 0; r < PF; ++r) { ka[r] = keys.at(toff, r * G::T * 8); kb[r] = keys.at(toff, (CS + r * G::T) * 8); }
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const double a = ka[r % PF], b = kb[r % PF];
        if (r + PF < G::E) { ka[r % PF] = keys.at(toff, (r + PF) * G::T * 8); kb[r % PF] = keys.at(toff, (CS + (r + PF) * G::T) * 8); }
        const double x = v[r];
        v[r] = nxt.at(toff, G::idxA(r, 0) * 8);
        if constexpr (FOLD) {
            acc0[r] = hxf::mac_fold(acc0[r], x, a, m);
            acc1[r] = hxf::mac_fold(acc1[r], x, b, m);
        } else {
            acc0[r] = hxf::reduce(acc0[r] + hxf::mul_mod(x, a, m), m);
            acc1[r] = hxf::reduce(acc1[r] + hxf::mul_mod(x, b, m), m);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The FIRST multiply-accumulate of a workgroup (the d == i term): the accumulators are not live yet, so all 2 E key words
// are requested straight into their registers, right behind the E words of t_i and before anything is waited for -- one
// memory latency for the whole phase. (Through the three-deep ring this phase took 22 k cycles, twice a later round's:
// all 16 waves are in it at the same time, nobody has transform work to cover the ring's short reach.)
// LAZYFOLD (lazy kernels with the folded multiply-accumulate): nothing is range-reduced here -- t_i < q_i as it comes
// (k_ksx_intt has checked these very words), |t_i . key mod p| <= (0.5 + 0.378) p = 0.88p by mul_mod's general bound, which
// mac_fold accepts as an accumulator (|acc| <= 1.6p).
template <class G, bool LAZYFOLD = false, int CSW = 0>
__device__ __forceinline__ void mac_keys_first(double (&acc0)[G::E], double (&acc1)[G::E], double (&v)[G::E],
                                               const u64* __restrict__ t, const double* __restrict__ k0,
                                               const double* __restrict__ next, int tid, const Mod m) {
    static_assert(G::KL <= 2, "direct B-order loads");
    constexpr int CS = CSW ? CSW : G::N;
    const RowStream<double> keys(k0, 2 * CS * 8), nxt(next, G::N * 8);
    const u32 toff = u32(tid) * 8, tB = u32(G::idxB(0, tid));
    u64 raw[G::E];
#pragma unroll
    for (int r = 0; r < G::E; ++r) raw[r] = (t + G::idxB(r, 0))[tB];
#pragma unroll
    for (int r = 0; r < G::E; ++r) { acc0[r] = keys.at(toff, r * G::T * 8); acc1[r] = keys.at(toff, (CS + r * G::T) * 8); }
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        v[r] = nxt.at(toff, G::idxA(r, 0) * 8);
        if constexpr (LAZYFOLD) {
            const double x = hxf::to_f64_lt52(raw[r]);
            acc0[r] = hxf::mul_mod(x, acc0[r], m);
            acc1[r] = hxf::mul_mod(x, acc1[r], m);
        } else {
            const double x = hxf::reduce(hxf::to_f64(raw[r]), m);
            acc0[r] = hxf::reduce(hxf::mul_mod(x, acc0[r], m), m);
            acc1[r] = hxf::reduce(hxf::mul_mod(x, acc1[r], m), m);
        }
    }
}

// (x . y) mod p of two natural-order limbs as centred doubles in B register order (fused multiply + relinearize). In-range
// operands: |x|, |y| <= p/2 after centring, |x.y mod p| <= 0.7p. Lanes that own at most four adjacent words load straight at the B
// positions; otherwise both operands are read in A order (the product is element-wise: any common order will do) and the
// product takes the cross-wave re-deal of load_natural_to_B.
template <class G>
__device__ __forceinline__ void load_product_to_B(double (&v)[G::E], const u64* __restrict__ x, const u64* __restrict__ y,
                                                  double* lds, int tid, const Mod m) {
    if constexpr (G::KL <= 2) {
        const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
        for (int r = 0; r < G::E; ++r)
            v[r] = hxf::reduce(hxf::mul_mod(hxf::reduce(hxf::to_f64((x + G::idxB(r, 0))[tB]), m),
                                            hxf::reduce(hxf::to_f64((y + G::idxB(r, 0))[tB]), m), m), m);
    } else {
#pragma unroll
        for (int r = 0; r < G::E; ++r)
            v[r] = hxf::reduce(hxf::mul_mod(hxf::reduce(hxf::to_f64((x + G::idxA(r, 0))[u32(tid)]), m),
                                            hxf::reduce(hxf::to_f64((y + G::idxA(r, 0))[u32(tid)]), m), m), m);
        redeal_x<G, false, true>(v, lds, tid, [](int r, int t) { return G::idxA(r, t); },
                                 [](int r, int t) { return G::idxB(r, t); });
    }
}

// position of register r / thread tid in a natural-order array about to enter an INVERSE transform (B order): direct
// for small lane runs, A order (then re-dealt through LDS) otherwise -- see load_natural_to_B
template <class G>
__device__ __forceinline__ int in_pos(int r, int tid) { return G::KL <= 2 ? G::idxB(r, tid) : G::idxA(r, tid); }

// ---- special slot: steps 1-4 for one instance -------------------------------------------------------------------
// step 4 for one k. The reference forms s'_k = INTT_{q_sp}(prod[k][special]) + floor(q_sp/2) (mod q_sp), canonical
// (intt2_redu.hpp:25,43), and every limb i then uses (s'_k + fix_i) mod q_i with fix_i = -floor(q_sp/2) mod q_i
// (intt2_redu.hpp:31-32, 49-51) -- that is, y_k = s'_k - floor(q_sp/2), the EXACT centred remainder of the inverse transform's
// output in [-floor(q_sp/2), floor(q_sp/2)]. The scratch holds y_k (a signed integer in a double): k_ksx_main needs neither
// the addition nor, when the moduli are of one size, a range reduction (|y_k| <= 0.5 rho q_i).
// (`gate`: the second of the two inverse transforms follows the first one's cross-wave reads -- ntt_core.hpp ReadersGate)
template <class G, class W>
__device__ __forceinline__ void ksx_special_down(double (&v)[G::E], double* __restrict__ dst, double* lds, int tid,
                                                 const double* ts, const KsModF64& msp, ReadersGate<G>& gate) {
    W::template inverse<false, typename W::NoHook, false, ReadersGate<G>>(v, lds, tid, ts + 2 * G::N, ts + 3 * G::N, msp.m, msp.sc,
                                                                         typename W::NoHook(), 0u, &gate);
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const double c = hxf::lift(v[r], msp.m);                   // canonical [0, q_sp)
        (dst + G::idxA(r, 0))[u32(tid)] = c > msp.half ? c - msp.m.p : c;
    }
}

// step 1: c_d = INTT_{q_d}(t_target[d]) as canonical doubles in natural order, one workgroup per (instance, limb).
// (Kept out of k_ksx_special: an inverse transform beside the 64 accumulator registers made the allocator spill a few
// accumulators, and every reload inside the multiply-accumulate waits for the whole key prefetch queue -- vector
// memory returns in order. That version spent 42 k cycles per multiply-accumulate instead of 10 k.)
template <int LOGN, int LOGE, int LAZY, bool FUSED = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE), KX_WAVES(LOGE)) void k_ksx_intt(KsArgsX a) {
    using G = Geom<LOGN, LOGE>;
    using W = WgNttF64<LOGN, LOGE, LAZY, 0, 0, true, HX_FWD_PRIO, 0, false, -1, KX_ISCHED != 0>;            // inverse without the w/p table
    extern __shared__ __attribute__((aligned(16))) double ldsx[];
    const XcdWalk wk = xcd_walk(a.nb * a.nsel);
    if (wk.pos >= wk.end) return;
    ReadersGate<G> gate(ldsx);                                    // every transform of the item loop follows an inverse one (ntt_core.hpp)
    // the next item's words are requested into spare registers behind the last per-lane twiddle request of the current
    // transform (WgNttF64::inverse's `before_uniform` hook; see k_ntt_inv_p): the item loop never waits for its input
    auto src_of = [&](u32 item) -> const u64* {
        if constexpr (FUSED) return nullptr;
        else return a.t_target + size_t(KX_ALIASED(4, item / a.nsel) * a.L + sel_limb(a, item % a.nsel)) * G::N;
    };
    u64 raw[G::E];
    hxf::RangeMask bad = 0;                                             // a t_target word >= its modulus (FP64 precondition)
    if constexpr (!FUSED && G::KL <= 2) {
        const u32 tB = u32(G::idxB(0, int(threadIdx.x)));
        const u64* p0 = src_of(wk.pos);
#pragma unroll
        for (int r = 0; r < G::E; ++r) raw[r] = (p0 + G::idxB(r, 0))[tB];
    }
#pragma unroll 1
    for (u32 item = wk.pos; item < wk.end; item += wk.step) {     // item = b*nsel + (number of d among this launch's limbs)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u32 ib = item / a.nsel;
        const u32 d = __builtin_amdgcn_readfirstlane(sel_limb(a, item - ib * a.nsel));
        const u32 row = ib * a.L + d;                                 // b*L + d
        const KsModF64 md = a.mods[d];
        u32 toff = KX_ALIASED(16, d) * 4 * G::N;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        double v[G::E];
        if constexpr (FUSED) {                                    // t_target[d] = a_1[d] . b_1[d]
            const size_t at = ((size_t(ib) * 2 + 1) * a.L + d) * G::N;
            load_product_to_B<G>(v, a.mul_a + at, a.mul_b + at, ldsx, tid, md.m);
            W::template inverse<false, typename W::NoHook, false, ReadersGate<G>>(v, ldsx, tid, tb + 2 * G::N, tb + 3 * G::N, md.m, md.sc, typename W::NoHook(), 0u, &gate);
        } else if constexpr (G::KL <= 2) {
            // canonical words as they are: the first inverse stage takes X + Y < 2p and |X - Y| < p (f64_arith.hpp)
            const u64 qd = (u64)md.m.p;
#pragma unroll
            for (int r = 0; r < G::E; ++r) v[r] = hxf::to_f64_lt52_checked(raw[r], qd, bad);
            const u32 nitem = item + wk.step < wk.end ? item + wk.step : item;       // (last round: a harmless re-read)
            const u64* pn = src_of(nitem);
            const u32 tB = u32(G::idxB(0, tid));
            auto request_next = [&] {
#pragma unroll
                for (int r = 0; r < G::E; ++r) raw[r] = (pn + G::idxB(r, 0))[tB];
            };
            W::template inverse<false, decltype(request_next), (KX_IPRE != 0), ReadersGate<G>>(v, ldsx, tid, tb + 2 * G::N, tb + 3 * G::N, md.m, md.sc, request_next, 0u, &gate);
        } else {
            load_natural_to_B<G>(v, a.t_target + size_t(row) * G::N, ldsx, tid, md.m, bad);
            W::template inverse<false, typename W::NoHook, false, ReadersGate<G>>(v, ldsx, tid, tb + 2 * G::N, tb + 3 * G::N, md.m, md.sc, typename W::NoHook(), 0u, &gate);
        }
        double* cd = a.c + size_t(row) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) (cd + G::idxA(r, 0))[u32(tid)] = hxf::lift(v[r], md.m);
    }
    hxf::report_range(bad, a.range_flag);
}

// steps 2-4 for the special slot of one instance: acc_k = sum_d NTT_{q_sp}(c_d mod q_sp) . key[d][special][k], then
// s'_k = INTT_{q_sp}(acc_k) + floor(q_sp/2)
// SKIP (lazy kernels, moduli within LAZY_SKIP_MAX_RATIO of each other): c_d enters the transform as it is, on the shifted schedule
template <int LOGN, int LOGE, int LAZY, bool SKIP = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE), KX_WAVES(LOGE)) void k_ksx_special(KsArgsX a) {
    using G = Geom<LOGN, LOGE>;
    static_assert(!SKIP || LAZY != 0, "SKIP is a lazy-kernel variant");
    using W = WgNttF64<LOGN, LOGE, LAZY, KX_PRE, SKIP ? 1 : 0, true, HX_FWD_PRIO, 0, KX_SEMIU_ON(LAZY), KX_XSCHED ? 0 : -1, KX_ISCHED != 0>;   // forward output -> mac_fold
    extern __shared__ __attribute__((aligned(16))) double ldsx[];
    const u32 L = a.L;
    const u32 isp = a.K - 1;
    const KsModF64 msp = a.mods[isp];
    const XcdWalk wk = xcd_walk(a.nb);
    ReadersGate<G> gate(ldsx);
#pragma unroll 1
    for (u32 b = wk.pos; b < wk.end; b += wk.step) {
    double acc0[G::E], acc1[G::E];
    double v[G::E];                                               // between rounds: the next round's input, A order
#ifdef KX_TIMELINE
    const u32 kx_slot = b;
#endif
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const double* c0 = a.c + size_t(KX_ALIASED(2, b)) * L * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; v[r] = (c0 + G::idxA(r, 0))[u32(tid)]; }
    }
    // (Walking the limbs in a different order per instance -- the sum is exact, its order free -- does not shorten this
    // kernel's multiply-accumulate, 12-13 k cycles against 5 k in k_ksx_main: its workgroups start together and stay in
    // step, so all 16 waves of a CU are in that phase at once and nobody has transform work to cover the key latency.
    // Round 6: neither does a start skew of every other workgroup by 8 k / 16 k cycles, nor a key ring of 4 or 6 pairs in
    // this kernel only -- 217.6-218.0 k keyswitch/s shipped against 217.0-217.8 k for all four, same box:
    // tools/experiments/r06_special_dephase.patch.)
#pragma unroll 1
    for (u32 it = 0; it < L; ++it) {
        int tid = threadIdx.x;                                    // laundered per round (see k_ksf_up)
        asm volatile("" : "+v"(tid));
        u32 tsp = KX_ALIASED(16, isp) * 4 * G::N;
        asm volatile("" : "+s"(tsp));
        const double* ts = a.tables + tsp;
        KX_STAMP(4 * it + 0);
        if constexpr (!SKIP) {
#pragma unroll
            for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(v[r], msp.m);             // intt1_redu.hpp:36-42
        }
        KX_STAMP(4 * it + 1);
        const double* k0 = key_row<G>(a, it, L);
        const u32 nd = it + 1 < L ? it + 1 : it;                  // (the last limb is requested twice: harmless)
        W::template forward<false, false>(v, ldsx, tid, ts, ts + G::N, msp.m);
        KX_STAMP(4 * it + 2);
        mac_keys<G, true>(acc0, acc1, v, k0, a.c + (size_t(KX_ALIASED(2, b)) * L + nd) * G::N, tid, msp.m);
    }
    // (the folded multiply-accumulate leaves |acc| <= 1.7p: one reduction in front of the inverse transforms)
#pragma unroll
    for (int r = 0; r < G::E; ++r) { acc0[r] = hxf::reduce(acc0[r], msp.m); acc1[r] = hxf::reduce(acc1[r], msp.m); }
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 tsp = KX_ALIASED(16, isp) * 4 * G::N;
        asm volatile("" : "+s"(tsp));
        KX_STAMP(4 * L + 0);
        ksx_special_down<G, W>(acc0, a.s + (size_t(b) * 2 + 0) * G::N, ldsx, tid, a.tables + tsp, msp, gate);
    }
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 tsp = KX_ALIASED(16, isp) * 4 * G::N;
        asm volatile("" : "+s"(tsp));
        KX_STAMP(4 * L + 4);
        ksx_special_down<G, W>(acc1, a.s + (size_t(b) * 2 + 1) * G::N, ldsx, tid, a.tables + tsp, msp, gate);
        KX_STAMP(4 * L + 8);
    }
    }
}

// ---- decomposition slots: steps 2-3 and 5-7 for one (instance, limb) ----------------------------------------------
// steps 5-7 for one k: w = NTT((s'_k + fix_i) mod q_i) from the raw s'_k words in v (A order); result[k][i] += (acc - w) * msf_i
// FUSED (k = 0, 1): the "old result" is component k of the ciphertext product, formed here from the operand limbs
// (a0, a1, b0, b1 of this limb, natural order): k = 0: a0 b0; k = 1: a0 b1 + a1 b0; `res` is written, not accumulated into
// v holds y_k = s'_k - floor(q_sp/2) (ksx_special_down), which IS (s'_k + fix_i) mod q_i of intt2_redu.hpp:49-51 up to a
// multiple of q_i. SKIP: |y_k| <= 0.5 rho q_i goes into the transform as it is (standard schedule: 0.625 -> 3.77 after three
// stages at rho = 1.25). The accumulators arrive un-reduced from mac_fold in the lazy kernels (|acc| <= 1.7p):
// |acc - w| <= 1.7p + 2.14p = 3.84p < 2^53 = 3.97p, and mul_shoup of that is exact (|h| < 2^103, |h - k p| <= 1.5p).
// PRED (the N = 32768 kernels): v arrives ready for the sub-transform (already reduced / combined across the halves); `top` = which half
template <class G, class W, int FUSED_K = -1, bool SKIP = false, bool PRED = false>
__device__ __forceinline__ void ksx_down_round(double (&v)[G::E], const double (&acc)[G::E], u64* __restrict__ res,
                                               double* lds, int tid, const double* tb, const KsModF64& md, hxf::RangeMask& bad,
                                               const u64* a0 = nullptr, const u64* a1 = nullptr, const u64* b0 = nullptr,
                                               const u64* b1 = nullptr, u32 top = 0) {
    const Mod m = md.m;
    if constexpr (!SKIP && !PRED) {
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(v[r], m);
    }
    // (the w/p table -- read by the strict kernels' semi-strict passes only -- lies one FULL transform's worth of words behind w)
    W::template forward<false, false>(v, lds, tid, tb, tb + (PRED ? 2 : 1) * G::N, m, typename W::NoHook(), typename W::NoHook(), top);   // |w| <= 2.14p
#pragma unroll
    for (int r = 0; r < G::E; ++r) v[r] = hxf::mul_shoup(acc[r] - v[r], md.msf, md.msf_p, m);   // ms.hpp:70-82
    if constexpr (FUSED_K >= 0 && G::KL <= 2) {
        const u32 tB = u32(G::idxB(0, tid));
        auto ld = [&](const u64* p, int r) { return hxf::reduce(hxf::to_f64((p + G::idxB(r, 0))[tB]), m); };
#pragma unroll
        for (int r0 = 0; r0 < G::E; r0 += 4) {                    // four coefficients at a time: 8 or 16 loads in flight
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r) {
                double old;
                if constexpr (FUSED_K == 0) old = hxf::mul_mod(ld(a0, r), ld(b0, r), m);
                else old = hxf::reduce(hxf::mul_mod(ld(a0, r), ld(b1, r), m) + hxf::mul_mod(ld(a1, r), ld(b0, r), m), m);
                (res + G::idxB(r, 0))[tB] = hxf::from_f64(hxf::lift(hxf::reduce(old + v[r], m), m));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    if constexpr (G::KL <= 2) {
        // read-modify-write at the B positions, half of the registers at a time (a full register copy of the old words
        // does not fit beside the accumulators)
        const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
        for (int r0 = 0; r0 < G::E; r0 += G::E / 2) {
            u64 old[G::E / 2];
            const u64 qi = (u64)m.p;
#pragma unroll
            for (int r = 0; r < G::E / 2; ++r) old[r] = (res + G::idxB(r0 + r, 0))[tB];
#pragma unroll
            for (int r = 0; r < G::E / 2; ++r) {
                const double rr = hxf::reduce(hxf::to_f64_lt52_checked(old[r], qi, bad) + v[r0 + r], m);   // fpga.cpp:453-457
                (res + G::idxB(r0 + r, 0))[tB] = hxf::from_f64(hxf::lift(rr, m));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // result is natural order, v is B order: the old words go through LDS (written at their A positions, read and
    // replaced by the sums at the thread's own B positions, read back at the A positions), eight loads in flight at a
    // time -- a register copy of them would not fit beside the accumulators
    double* const atA = lds + G::pad(G::idxA(0, tid));
    double* const atB = lds + G::pad(G::idxB(0, tid));
    __syncthreads();                                              // other waves may still read the last re-deal
    // the value the output is added to, at A position r: the old result word, or (fused) component k of the ciphertext product
    auto old_at_A = [&](int r) {
        auto ld = [&](const u64* p) { return hxf::reduce(hxf::to_f64((p + G::idxA(r, 0))[u32(tid)]), m); };
        if constexpr (FUSED_K == 0) return hxf::mul_mod(ld(a0), ld(b0), m);
        else if constexpr (FUSED_K == 1) return hxf::reduce(hxf::mul_mod(ld(a0), ld(b1), m) + hxf::mul_mod(ld(a1), ld(b0), m), m);
        else return hxf::reduce(hxf::to_f64_checked((res + G::idxA(r, 0))[u32(tid)], m, bad), m);
    };
#pragma unroll
    for (int r0 = 0; r0 < G::E; r0 += (FUSED_K == 1 ? 4 : 8)) {   // 8 (or 16: four operand streams) loads in flight at a time
#pragma unroll
        for (int r = r0; r < r0 + (FUSED_K == 1 ? 4 : 8); ++r) atA[G::pad(G::idxA(r, 0))] = old_at_A(r);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
        const double o = atB[G::pad(G::idxB(r, 0))];
        atB[G::pad(G::idxB(r, 0))] = hxf::lift(hxf::reduce(o + v[r], m), m);         // fpga.cpp:453-457
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < G::E; ++r) (res + G::idxA(r, 0))[u32(tid)] = hxf::from_f64(atA[G::pad(G::idxA(r, 0))]);
}

template <int LOGN, int LOGE, int LAZY, bool FUSED = false, bool SKIP = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE), KX_WAVES(LOGE)) void k_ksx_main(KsArgsX a) {
    using G = Geom<LOGN, LOGE>;
    static_assert(!SKIP || LAZY != 0, "SKIP is a lazy-kernel variant");
    using W = WgNttF64<LOGN, LOGE, LAZY, KX_PRE, 0, false, HX_FWD_PRIO, 0, KX_SEMIU_ON(LAZY), KX_XSCHED ? 1 : -1>;               // mod-down transforms: centred input, tail -> acc - w
    using WU = WgNttF64<LOGN, LOGE, LAZY, KX_PRE, SKIP ? 1 : 0, false, HX_FWD_PRIO, 0, KX_SEMIU_ON(LAZY), KX_XSCHED ? 0 : -1>;   // mod-up transforms (SKIP: canonical c_d as it is), tail -> mac_fold
    // lazy kernels: the d == i term and the accumulators go un-reduced into the folded multiply-accumulate (f64_arith.hpp mac_fold,
    // |acc| <= 1.6p between rounds). The strict kernels (moduli up to 2^52) fold theirs too (transform output |x| <= p/2 + 2,
    // accumulators <= 0.9p between terms) and reduce the accumulators once in front of the mod-down, whose epilogue needs them centred
    // at this modulus size; the d == i term keeps its reduced form there
    constexpr bool LAZYFOLD = LAZY != 0;
    extern __shared__ __attribute__((aligned(16))) double ldsx[];
    const u32 L = a.L;
    // one item per workgroup: as a persistent loop (xcd_walk) this kernel measured 2.6-10 % slower in rounds 2-4 (what the compiler hoists
    // out of the item loop costs more registers than the dispatch gaps cost time; tools/experiments/persistent_main_out_of_line.patch)
    {
    const u32 item = __builtin_amdgcn_readfirstlane(xcd_item_x(blockIdx.x, gridDim.x));
#ifdef KX_TIMELINE
    const u32 kx_slot = item;
#endif
    // instance-major, XCD-contiguous: the workgroups (one per limb of this launch) that read the same c_d and s' run side by side on one
    // XCD (slot-major -- an XCD's keys L2-resident, c from the Infinity Cache -- measured slower: 234 against 215 us per generation)
    const u32 b = item / a.nsel, i = sel_limb(a, item - b * a.nsel);
    // (through the constant address space: inside an item loop that also stores to global memory a plain read of these
    // wave-uniform constants becomes a VECTOR load, and p, 1/p, msf ... then occupy vector registers the accumulators need)
    const KsModF64 md = load_mod_const(a.mods + i);
    const Mod m = md.m;
    // round `it` reads c_it (it < L, skipping it == i) or s'_{it-L}
    const u32 bc = KX_ALIASED(2, b), bt = KX_ALIASED(4, b), br = KX_ALIASED(8, b);
    auto round_src = [&](u32 it) { return it < L ? a.c + (size_t(bc) * L + it) * G::N : a.s + (size_t(bc) * 2 + (it - L)) * G::N; };
    const u32 first = i == 0 ? 1u : 0u;
    double acc0[G::E], acc1[G::E];
    double v[G::E];                                               // between rounds: the next round's input, A order
    {
        // d == i: NTT_{q_i}(INTT_{q_i}(t_i) mod q_i) = t_i (the reference recomputes it; same value for in-range data)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const double* k0 = key_row<G>(a, i, i);
        KX_STAMP(60);
        if constexpr (!FUSED && G::KL <= 2) {
            KX_STAMP(61);
            mac_keys_first<G, LAZYFOLD>(acc0, acc1, v, a.t_target + (size_t(bt) * L + i) * G::N, k0, round_src(first), tid, m);
        } else {
#pragma unroll
            for (int r = 0; r < G::E; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
            if constexpr (FUSED) {
                const size_t at = ((size_t(b) * 2 + 1) * L + i) * G::N;
                load_product_to_B<G>(v, a.mul_a + at, a.mul_b + at, ldsx, tid, m);
            } else {
                hxf::RangeMask ignore = 0;                              // (k_ksx_intt has checked this limb)
                load_natural_to_B<G>(v, a.t_target + (size_t(bt) * L + i) * G::N, ldsx, tid, m, ignore);
            }
            KX_STAMP(61);
            mac_keys<G>(acc0, acc1, v, k0, round_src(first), tid, m);
        }
    }
    // rounds d != i: acc += NTT(c_d mod q_i) . key[d][i]
#pragma unroll 1
    for (u32 it = first; it < L;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 toff = KX_ALIASED(16, i) * 4 * G::N;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        KX_STAMP(4 * it + 0);
        if constexpr (!SKIP) {
#pragma unroll
            for (int r = 0; r < G::E; ++r) v[r] = hxf::reduce(v[r], m);             // intt1_redu.hpp:36-42
        }
        KX_STAMP(4 * it + 1);
        u32 nit = it + 1;
        if (nit == i) ++nit;
        const double* k0 = key_row<G>(a, it, i);
        WU::template forward<false, false>(v, ldsx, tid, tb, tb + G::N, m);         // |u| <= 2.14p (SKIP: 3.45p)
        KX_STAMP(4 * it + 2);
        mac_keys<G, true>(acc0, acc1, v, k0, round_src(nit), tid, m);               // nit <= L: s'_0 follows the last c_d
        it = nit;
    }
    // (lazy kernels: the accumulators stay as mac_fold leaves them, |acc| <= 1.7p -- ksx_down_round)
    if constexpr (LAZY == 0) {
#pragma unroll
        for (int r = 0; r < G::E; ++r) { acc0[r] = hxf::reduce(acc0[r], m); acc1[r] = hxf::reduce(acc1[r], m); }
    }
    // rounds L, L+1 (k = 0, 1)
    hxf::RangeMask bad = 0;                                             // a result word >= its modulus (FP64 precondition)
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 toff = KX_ALIASED(16, i) * 4 * G::N;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        KX_STAMP(4 * L + 0);
        const size_t o0 = ((size_t(br) * 2 + 0) * L + i) * G::N, o1 = ((size_t(br) * 2 + 1) * L + i) * G::N;
        if constexpr (FUSED) ksx_down_round<G, W, 0, SKIP>(v, acc0, a.result + o0, ldsx, tid, tb, md, bad, a.mul_a + o0, a.mul_a + o1, a.mul_b + o0, a.mul_b + o1);
        else ksx_down_round<G, W, -1, SKIP>(v, acc0, a.result + o0, ldsx, tid, tb, md, bad);
        const double* nxt = a.s + (size_t(bc) * 2 + 1) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = (nxt + G::idxA(r, 0))[u32(tid)];
        KX_STAMP(4 * L + 4);
    }
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 toff = KX_ALIASED(16, i) * 4 * G::N;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        const size_t o0 = ((size_t(br) * 2 + 0) * L + i) * G::N, o1 = ((size_t(br) * 2 + 1) * L + i) * G::N;
        if constexpr (FUSED) ksx_down_round<G, W, 1, SKIP>(v, acc1, a.result + o1, ldsx, tid, tb, md, bad, a.mul_a + o0, a.mul_a + o1, a.mul_b + o0, a.mul_b + o1);
        else ksx_down_round<G, W, -1, SKIP>(v, acc1, a.result + o1, ldsx, tid, tb, md, bad);
        KX_STAMP(4 * L + 8);
        hxf::report_range(bad, a.range_flag);
    }
    }
}

// =====================================================================================================================================
// N = 32768 on the slot-major pipeline (round 5; beyond the reference's envelope). 64 registers of polynomial + 128 of
// accumulators do not fit a 1024-thread workgroup, so every 32768-point transform is cut in TWO, the way keyswitch_lat.hip
// cuts a 16384-point one in four: its outermost stage is a radix-2 step across the two halves of the polynomial, the other fourteen are
// two independent 16384-point sub-transforms (WgNttF64<14, 4, ..., TOP = 1>: stage numbers, reduction schedule and twiddle indices of
// the full transform, the half number on top of every group index). A workgroup = one (instance, limb, HALF) with the geometry, the
// registers and the multiply-accumulate of the N = 16384 kernels:
//   k_ksh_intt    (b, d, h)   fourteen inverse stages on half h of t_target[d] (NTT-domain block h)               -> csub (raw)
//   k_ksh_finish              the inverse's last stage (n^-1 folded in) across the two halves, elementwise         -> c / y
//   k_ksh_special (b, h)      per d: c_d's halves are combined ON LOAD by the forward transform's first stage (the workgroup reads
//                             both, keeps the half it owns), fourteen forward stages, multiply-accumulate with its half of the key
//                             rows; then fourteen inverse stages on its half of the accumulators                     -> ssub (raw)
//   k_ksh_main    (b, i, h)   the same mod-up, the d == i term from its block of t_target[i], the two mod-down rounds on its
//                             block of result[k][i]
// The NTT-domain index space splits into contiguous blocks (sub-transform h produces / consumes block h), so t_target, the keys and
// result need no exchange at all; only the coefficient-domain arrays (c_d, s') are read in full by both halves: + 1 load, a reduction and
// 7 FP64 operations per coefficient and round. Same arithmetic as the monolithic transforms of the (b, d)-major kernels: bit-identical.
constexpr int KSH_HB = 8;      // words of the other half requested at a time in ksh_combine
template <class G, int LAZY, bool SKIP, int SHIFT, unsigned XS = 0u>
__device__ __forceinline__ void ksh_combine(double (&v)[G::E], const double* __restrict__ hi_row, int tid, const double* w, const Mod m, u32 h) {
    // forward global stage 1 of the 2^15-point transform (one twiddle: index 1), this workgroup keeps output half h
    const double W1 = ((ctw_t)w)[1];
    // (the transform's own schedule; on an X schedule -- XS = the sub-transforms' mask -- stage 1 is never a reduction point)
    constexpr bool red = XS ? false : (LAZY == 0 || hxf::lazy_fwd_reduce_after(1, 15, LAZY ? LAZY : 3, SHIFT));
    // the other half's words, HB at a time: all sixteen in flight at once need 32 registers the accumulators do not leave
    constexpr int HB = KSH_HB;
#pragma unroll
    for (int r0 = 0; r0 < G::E; r0 += HB) {
        double hi[HB];
#pragma unroll
        for (int r = 0; r < HB; ++r) hi[r] = (hi_row + G::idxA(r0 + r, 0))[u32(tid)];
#pragma unroll
        for (int r = 0; r < HB; ++r) {
            double lo = v[r0 + r], y = hi[r];
            if constexpr (!SKIP) { lo = hxf::reduce(lo, m); y = hxf::reduce(y, m); }    // intt1_redu.hpp:36-42 / intt2_redu.hpp:49-51
            const double t = hxf::mul_mod(y, W1, m);
            const double o = h ? lo - t : lo + t;
            v[r0 + r] = red ? hxf::reduce(o, m) : o;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// FUSED (hexl_multiply_relinearize at N = 32768): t_target[d] is the product a_1[d] . b_1[d] of the operand limbs, formed block by block (the
// product is element-wise in the NTT domain, so half h of it needs half h of the operands only); k_ksh_main forms the other two components
template <int LAZY, bool FUSED = false>
__global__ __launch_bounds__(1024, 4) void k_ksh_intt(KsArgsX a) {
    using G = Geom<14, 4>;
    using W = WgNttF64<14, 4, LAZY, 0, 0, true, HX_FWD_PRIO, 1>;
    constexpr u32 NF = 2 * G::N;
    extern __shared__ __attribute__((aligned(16))) double ldsx[];
    const XcdWalk wk = xcd_walk(a.nb * a.nsel * 2);
    hxf::RangeMask bad = 0;
    ReadersGate<G> gate(ldsx);                                    // inverse after inverse in one workgroup (ntt_core.hpp)
#pragma unroll 1
    for (u32 unit = wk.pos; unit < wk.end; unit += wk.step) {     // unit = (b * nsel + number of d among this launch's limbs) * 2 + h
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u32 h = unit & 1, item = unit >> 1, ib = item / a.nsel;
        const u32 d = __builtin_amdgcn_readfirstlane(sel_limb(a, item - ib * a.nsel));
        const u32 row = ib * a.L + d;
        const KsModF64 md = a.mods[d];
        u32 toff = d * 4 * NF;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        double v[G::E];
        if constexpr (FUSED) {
            const size_t at = ((size_t(ib) * 2 + 1) * a.L + d) * NF + h * G::N;
            load_product_to_B<G>(v, a.mul_a + at, a.mul_b + at, ldsx, tid, md.m);
        } else {
            const u64* src = a.t_target + size_t(row) * NF + h * G::N;
            const u64 qd = (u64)md.m.p;
            const u32 tB = u32(G::idxB(0, tid));
#pragma unroll
            for (int r = 0; r < G::E; ++r) v[r] = hxf::to_f64_lt52_checked((src + G::idxB(r, 0))[tB], qd, bad);    // canonical words as they are
        }
        W::template inverse<false, typename W::NoHook, false, ReadersGate<G>>(v, ldsx, tid, tb + 2 * NF, tb + 3 * NF, md.m, md.sc, typename W::NoHook(), h, &gate);
        double* dst = a.csub + (size_t(row) * 2 + h) * G::N;
#pragma unroll
        for (int r = 0; r < G::E; ++r) (dst + G::idxA(r, 0))[u32(tid)] = v[r];
    }
    hxf::report_range(bad, a.range_flag);
}

// the last inverse stage across the halves, n^-1 folded in (inv_stages_f64's fused last stage on register pairs), then what the
// consumers read: WHICH = 0: c_d canonical (rows b * L + d of this launch's limbs), 1: y_k = s'_k - floor(q_sp / 2), the exact centred
// remainder (ksx_special_down). Elementwise, HBM-bound, a few microseconds per chunk.
template <int WHICH>
__global__ __launch_bounds__(256) void k_ksh_finish(KsArgsX a, u32 rows) {
    constexpr u32 H = 1u << 14, NF = 2 * H;
    const u32 row = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows || j >= H) return;
    const u32 limb = WHICH == 0 ? row % a.L : a.K - 1;
    const KsModF64 md = a.mods[limb];
    const double* tb = a.tables + size_t(limb) * 4 * NF;
    const double* sub = (WHICH == 0 ? a.csub : a.ssub) + size_t(row) * NF;
    double x[2] = {sub[j], sub[H + j]};
    inv_stages_f64<2, 0, 1, 14, 15, true, 3, true, true>(x, 0u, tb + 2 * NF, tb + 3 * NF, md.m, md.sc);
    double* dst = (WHICH == 0 ? a.c : a.s) + size_t(row) * NF;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double c = hxf::lift(x[k], md.m);
        dst[k * H + j] = WHICH == 0 ? c : (c > md.half ? c - md.m.p : c);
    }
}

template <int LAZY, bool SKIP>
__global__ __launch_bounds__(1024, 4) void k_ksh_special(KsArgsX a) {
    using G = Geom<14, 4>;
    using W = WgNttF64<14, 4, LAZY, KX_PRE, SKIP ? 1 : 0, true, HX_FWD_PRIO, 1, KX_SEMIU_ON(LAZY), KX_XSCHED ? 0 : -1>;
    constexpr u32 NF = 2 * G::N;
    extern __shared__ __attribute__((aligned(16))) double ldsx[];
    const u32 L = a.L, isp = a.K - 1;
    const KsModF64 msp = a.mods[isp];
    const XcdWalk wk = xcd_walk(a.nb * 2);
    ReadersGate<G> gate(ldsx);                                    // the two inverse transforms of a unit follow each other (ntt_core.hpp)
#pragma unroll 1
    for (u32 unit = wk.pos; unit < wk.end; unit += wk.step) {
        const u32 h = unit & 1, b = unit >> 1;
        double acc0[G::E], acc1[G::E], v[G::E];
        {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const double* c0 = a.c + size_t(b) * L * NF;
#pragma unroll
            for (int r = 0; r < G::E; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; v[r] = (c0 + G::idxA(r, 0))[u32(tid)]; }
        }
#pragma unroll 1
        for (u32 it = 0; it < L; ++it) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            u32 tsp = isp * 4 * NF;
            asm volatile("" : "+s"(tsp));
            const double* ts = a.tables + tsp;
            ksh_combine<G, LAZY, SKIP, (SKIP ? 1 : 0), W::XS>(v, a.c + (size_t(b) * L + it) * NF + G::N, tid, ts, msp.m, h);
            const double* k0 = key_row<G>(a, it, L) + h * G::N;
            const u32 nd = it + 1 < L ? it + 1 : it;
            W::template forward<false, false>(v, ldsx, tid, ts, ts + NF, msp.m, typename W::NoHook(), typename W::NoHook(), h);
            mac_keys<G, true, int(NF)>(acc0, acc1, v, k0, a.c + (size_t(b) * L + nd) * NF, tid, msp.m);
        }
#pragma unroll
        for (int r = 0; r < G::E; ++r) { acc0[r] = hxf::reduce(acc0[r], msp.m); acc1[r] = hxf::reduce(acc1[r], msp.m); }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            u32 tsp = isp * 4 * NF;
            asm volatile("" : "+s"(tsp));
            const double* ts = a.tables + tsp;
            double (&acc)[G::E] = k == 0 ? acc0 : acc1;
            W::template inverse<false, typename W::NoHook, false, ReadersGate<G>>(acc, ldsx, tid, ts + 2 * NF, ts + 3 * NF, msp.m, msp.sc, typename W::NoHook(), h, &gate);
            double* dst = a.ssub + ((size_t(b) * 2 + k) * 2 + h) * G::N;
#pragma unroll
            for (int r = 0; r < G::E; ++r) (dst + G::idxA(r, 0))[u32(tid)] = acc[r];
        }
    }
}

template <int LAZY, bool SKIP, bool FUSED = false>
__global__ __launch_bounds__(1024, 4) void k_ksh_main(KsArgsX a) {
    using G = Geom<14, 4>;
    using W = WgNttF64<14, 4, LAZY, KX_PRE, 0, false, HX_FWD_PRIO, 1, KX_SEMIU_ON(LAZY), KX_XSCHED ? 1 : -1>;                 // mod-down transforms: centred input
    using WU = WgNttF64<14, 4, LAZY, KX_PRE, SKIP ? 1 : 0, false, HX_FWD_PRIO, 1, KX_SEMIU_ON(LAZY), KX_XSCHED ? 0 : -1>;     // mod-up transforms
    constexpr u32 NF = 2 * G::N;
    constexpr bool LAZYFOLD = LAZY != 0;
    extern __shared__ __attribute__((aligned(16))) double ldsx[];
    const u32 L = a.L;
    const u32 unit = __builtin_amdgcn_readfirstlane(xcd_item_x(blockIdx.x, gridDim.x));   // (b * nsel + limb number) * 2 + h, XCD-contiguous
    const u32 h = unit & 1, item = unit >> 1;
    const u32 b = item / a.nsel, i = sel_limb(a, item - b * a.nsel);
    const KsModF64 md = load_mod_const(a.mods + i);
    const Mod m = md.m;
    auto round_src = [&](u32 it) { return it < L ? a.c + (size_t(b) * L + it) * NF : a.s + (size_t(b) * 2 + (it - L)) * NF; };
    const u32 first = i == 0 ? 1u : 0u;
    double acc0[G::E], acc1[G::E], v[G::E];
    {
        // d == i: NTT_{q_i}(INTT_{q_i}(t_i) mod q_i) = t_i -- this workgroup's block of it
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if constexpr (FUSED) {
#pragma unroll
            for (int r = 0; r < G::E; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
            const size_t at = ((size_t(b) * 2 + 1) * L + i) * NF + h * G::N;
            load_product_to_B<G>(v, a.mul_a + at, a.mul_b + at, ldsx, tid, m);                    // |x| <= 0.7 p, centred
            mac_keys<G, false, int(NF)>(acc0, acc1, v, key_row<G>(a, i, i) + h * G::N, round_src(first), tid, m);
        } else {
            mac_keys_first<G, LAZYFOLD, int(NF)>(acc0, acc1, v, a.t_target + (size_t(b) * L + i) * NF + h * G::N, key_row<G>(a, i, i) + h * G::N,
                                                 round_src(first), tid, m);
        }
    }
#pragma unroll 1
    for (u32 it = first; it < L;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 toff = i * 4 * NF;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        ksh_combine<G, LAZY, SKIP, (SKIP ? 1 : 0), WU::XS>(v, round_src(it) + G::N, tid, tb, m, h);
        u32 nit = it + 1;
        if (nit == i) ++nit;
        const double* k0 = key_row<G>(a, it, i) + h * G::N;
        WU::template forward<false, false>(v, ldsx, tid, tb, tb + NF, m, typename WU::NoHook(), typename WU::NoHook(), h);
        mac_keys<G, true, int(NF)>(acc0, acc1, v, k0, round_src(nit), tid, m);           // nit <= L: s'_0 follows the last c_d
        it = nit;
    }
    if constexpr (LAZY == 0) {
#pragma unroll
        for (int r = 0; r < G::E; ++r) { acc0[r] = hxf::reduce(acc0[r], m); acc1[r] = hxf::reduce(acc1[r], m); }
    }
    hxf::RangeMask bad = 0;
    {   // k = 0: y_0 (SKIP: |y| <= 0.625 q_i as it is, standard schedule) -> first stage across its halves -> this workgroup's half
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 toff = i * 4 * NF;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        ksh_combine<G, LAZY, SKIP, 0, W::XS>(v, round_src(L) + G::N, tid, tb, m, h);
        const size_t o0 = ((size_t(b) * 2 + 0) * L + i) * NF + h * G::N, o1 = ((size_t(b) * 2 + 1) * L + i) * NF + h * G::N;
        if constexpr (FUSED) ksx_down_round<G, W, 0, SKIP, true>(v, acc0, a.result + o0, ldsx, tid, tb, md, bad, a.mul_a + o0, a.mul_a + o1, a.mul_b + o0, a.mul_b + o1, h);
        else ksx_down_round<G, W, -1, SKIP, true>(v, acc0, a.result + o0, ldsx, tid, tb, md, bad, nullptr, nullptr, nullptr, nullptr, h);
        const double* nxt = round_src(L + 1);
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = (nxt + G::idxA(r, 0))[u32(tid)];
    }
    {   // k = 1
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        u32 toff = i * 4 * NF;
        asm volatile("" : "+s"(toff));
        const double* tb = a.tables + toff;
        ksh_combine<G, LAZY, SKIP, 0, W::XS>(v, round_src(L + 1) + G::N, tid, tb, m, h);
        const size_t o0 = ((size_t(b) * 2 + 0) * L + i) * NF + h * G::N, o1 = ((size_t(b) * 2 + 1) * L + i) * NF + h * G::N;
        if constexpr (FUSED) ksx_down_round<G, W, 1, SKIP, true>(v, acc1, a.result + o1, ldsx, tid, tb, md, bad, a.mul_a + o0, a.mul_a + o1, a.mul_b + o0, a.mul_b + o1, h);
        else ksx_down_round<G, W, -1, SKIP, true>(v, acc1, a.result + o1, ldsx, tid, tb, md, bad, nullptr, nullptr, nullptr, nullptr, h);
    }
    hxf::report_range(bad, a.range_flag);
}

// ---------------------------------------------------------------------------------------------
template <class K>
static int set_lds_x(K kern, size_t bytes) {
    HX_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// one kernel of the pipeline for the limbs `a` selects: stage 1 = k_ksx_intt (step 1), 2 = k_ksx_special (steps 2-4, special slot),
// 4 = k_ksx_main (steps 2-3 and 5-7 of the decomposition slots)
template <int LOGN, int LOGE, int LAZY, bool FUSED = false, bool SKIP = false>
static int launch_stage_x(hexl_ks_plan* p, const KsArgsX& a, int stage) {
    using G = Geom<LOGN, LOGE>;
    static PerDeviceOnce once;
    if (int rc0 = once.run(p->ctx->device, [] {
            int rc = set_lds_x(k_ksx_special<LOGN, LOGE, LAZY, SKIP>, G::LDS_USED);
            if (!rc) rc = set_lds_x(k_ksx_intt<LOGN, LOGE, LAZY, FUSED>, G::LDS_USED);
            if (!rc) rc = set_lds_x(k_ksx_main<LOGN, LOGE, LAZY, FUSED, SKIP>, G::LDS_USED);
            return rc;
        }))
        return rc0;
    hipStream_t st = p->cur;
    // persistent grids: 8 x g workgroups, g = workgroups per XCD = one per CU unless there are fewer items
    // (HEXL_KSX_PERSIST=0: one workgroup per item)
    static const int persist = [] { const char* e = getenv("HEXL_KSX_PERSIST"); return e ? atoi(e) : 1; }();
    // workgroups one CU holds at once: 16 (or 8) waves of 64 threads
    constexpr u32 wg_per_cu = (KX_WAVES(LOGE) * 4 * 64) / G::T ? (KX_WAVES(LOGE) * 4 * 64) / G::T : 1;
    auto grid_for = [&](u32 items) {
        const u32 per_xcd = (items + 7) / 8, slots = (((u32)p->ctx->num_cu + 7) / 8) * wg_per_cu;
        return dim3(8 * (persist && per_xcd > slots ? slots : per_xcd));
    };
    if (stage == 1)
        hipLaunchKernelGGL((k_ksx_intt<LOGN, LOGE, LAZY, FUSED>), grid_for(a.nb * a.nsel), dim3(G::T), G::LDS_USED, st, a);
    if (stage == 2)
        hipLaunchKernelGGL((k_ksx_special<LOGN, LOGE, LAZY, SKIP>), grid_for(a.nb), dim3(G::T), G::LDS_USED, st, a);
    if (stage == 4)
        hipLaunchKernelGGL((k_ksx_main<LOGN, LOGE, LAZY, FUSED, SKIP>), dim3(a.nb * a.nsel), dim3(G::T), G::LDS_USED, st, a);
    return 0;
}

// the kernel variant of a (tier, SKIP) pair. N = 16384 has all four reduction periods; the smaller rings keep 0 / 3 (a shorter period
// is always valid: fewer kernel variants), and so does the fused multiply + relinearize
template <int LOGN, int LOGE, bool FUSED>
static int launch_stage_tier(hexl_ks_plan* p, const KsArgsX& a, int stage, int tier, bool skip) {
    if constexpr (LOGN == 14 && LOGE == 4 && !FUSED) {
        switch (tier * 2 + (skip && tier ? 1 : 0)) {
            case 25: return launch_stage_x<14, 4, 12, false, true>(p, a, stage);
            case 24: return launch_stage_x<14, 4, 12>(p, a, stage);
            case 13: return launch_stage_x<14, 4, 6, false, true>(p, a, stage);
            case 12: return launch_stage_x<14, 4, 6>(p, a, stage);
            case 7:  return launch_stage_x<14, 4, 3, false, true>(p, a, stage);
            case 6:  return launch_stage_x<14, 4, 3>(p, a, stage);
            default: return launch_stage_x<14, 4, 0>(p, a, stage);
        }
    } else {
        if (!tier) return launch_stage_x<LOGN, LOGE, 0, FUSED>(p, a, stage);
        return skip ? launch_stage_x<LOGN, LOGE, 3, FUSED, true>(p, a, stage) : launch_stage_x<LOGN, LOGE, 3, FUSED>(p, a, stage);
    }
}

// One chunk. Plans of ONE tier: three launches, every limb in each. Plans of mixed tiers (hexl_ks_plan::mixed; round 5): k_ksx_intt once
// for the strict limbs and once for the lazy ones (an inverse transform has those two forms only), k_ksx_special in the special prime's
// tier, k_ksx_main once per tier present among the decomposition limbs -- e.g. bridge-seal's chain 52,30,30,40,27,27,27
// (experimental/bridge-seal/tests/seal_test.sh:20): limb 0 on the strict kernels, limbs 1-5 and the special prime at period 12, where
// rounds 1-4 ran the strict 14-instruction butterflies on all seven because one of them is 52-bit.
template <int LOGN, int LOGE, bool FUSED = false>
static int run_chunk_x(hexl_ks_plan* p, KsArgsX a, int stage_mask, hipEvent_t* ev) {
    hipStream_t st = p->cur;
    const u32 L = a.L;
    const bool per_limb = p->mixed && LOGE == 4;
    auto tier_of = [&](u32 i) { return per_limb ? (int)p->tier[i] : p->f64_lazy; };
    // launch `stage` once per group of limbs that share key(tier)
    auto per_group = [&](int stage, auto key) -> int {
        bool done[16] = {};
        for (u32 i0 = 0; i0 < L; ++i0) {
            if (done[i0]) continue;
            const int t0 = tier_of(i0);
            a.nsel = 0; a.selmap = 0;
            for (u32 i = i0; i < L; ++i)
                if (!done[i] && key(tier_of(i)) == key(t0)) { done[i] = true; a.selmap |= (unsigned long long)i << (4 * a.nsel++); }
            if (int rc = launch_stage_tier<LOGN, LOGE, FUSED>(p, a, stage, key(t0), p->x_skip)) return rc;
        }
        return 0;
    };
    // timing stages: 1 = step 1 (inverse transforms), 2 = special slot (steps 2-4), 4 = decomposition slots (steps 2-3, 5-7)
    if (ev) HX_CHECK(hipEventRecord(ev[0], st));
    if (stage_mask & 1)
        if (int rc = per_group(1, [](int t) { return t ? 3 : 0; })) return rc;
    if (ev) HX_CHECK(hipEventRecord(ev[1], st));
    if (stage_mask & 2) {
        a.nsel = L; a.selmap = 0xFEDCBA9876543210ull;
        if (int rc = launch_stage_tier<LOGN, LOGE, FUSED>(p, a, 2, tier_of(a.K - 1), p->x_skip)) return rc;
    }
    if (ev) HX_CHECK(hipEventRecord(ev[2], st));
    if (stage_mask & 4)
        if (int rc = per_group(4, [](int t) { return (LOGN == 14 && LOGE == 4 && !FUSED) ? t : (t ? 3 : 0); })) return rc;
    if (ev) HX_CHECK(hipEventRecord(ev[3], st));
    return (int)hipGetLastError();
}

// one chunk at N = 32768 (k_ksh_* kernels: every transform as two 16384-point halves). Tiers: strict / period 3 only (a shorter period
// is always valid); limbs of different tiers get a launch per tier like run_chunk_x.
template <int LAZY, bool SKIP, bool FUSED = false>
static int launch_stage_h(hexl_ks_plan* p, const KsArgsX& a, int stage) {
    using G = Geom<14, 4>;
    static PerDeviceOnce once;
    if (int rc0 = once.run(p->ctx->device, [] {
            int rc = set_lds_x(k_ksh_intt<LAZY, FUSED>, G::LDS_USED);
            if (!rc) rc = set_lds_x(k_ksh_special<LAZY, SKIP>, G::LDS_USED);
            if (!rc) rc = set_lds_x(k_ksh_main<LAZY, SKIP, FUSED>, G::LDS_USED);
            return rc;
        }))
        return rc0;
    hipStream_t st = p->cur;
    auto grid_for = [&](u32 items) {
        const u32 per_xcd = (items + 7) / 8, slots = ((u32)p->ctx->num_cu + 7) / 8;
        return dim3(8 * (per_xcd > slots ? slots : per_xcd));
    };
    if (stage == 1) hipLaunchKernelGGL((k_ksh_intt<LAZY, FUSED>), grid_for(a.nb * a.nsel * 2), dim3(G::T), G::LDS_USED, st, a);
    if (stage == 2) hipLaunchKernelGGL((k_ksh_special<LAZY, SKIP>), grid_for(a.nb * 2), dim3(G::T), G::LDS_USED, st, a);
    if (stage == 4) hipLaunchKernelGGL((k_ksh_main<LAZY, SKIP, FUSED>), dim3(a.nb * a.nsel * 2), dim3(G::T), G::LDS_USED, st, a);
    return 0;
}
template <bool FUSED>
static int launch_stage_h_tier(hexl_ks_plan* p, const KsArgsX& a, int stage, int tier, bool skip) {
    if (!tier) return launch_stage_h<0, false, FUSED>(p, a, stage);
    return skip ? launch_stage_h<3, true, FUSED>(p, a, stage) : launch_stage_h<3, false, FUSED>(p, a, stage);
}
template <bool FUSED = false>
static int run_chunk_h(hexl_ks_plan* p, KsArgsX a, int stage_mask, hipEvent_t* ev) {
    hipStream_t st = p->cur;
    const u32 L = a.L;
    auto tier_of = [&](u32 i) { return p->mixed ? (p->tier[i] ? 3 : 0) : (p->f64_lazy ? 3 : 0); };
    auto per_group = [&](int stage) -> int {
        bool done[16] = {};
        for (u32 i0 = 0; i0 < L; ++i0) {
            if (done[i0]) continue;
            a.nsel = 0; a.selmap = 0;
            for (u32 i = i0; i < L; ++i)
                if (!done[i] && tier_of(i) == tier_of(i0)) { done[i] = true; a.selmap |= (unsigned long long)i << (4 * a.nsel++); }
            if (int rc = launch_stage_h_tier<FUSED>(p, a, stage, tier_of(i0), p->x_skip)) return rc;
        }
        return 0;
    };
    if (ev) HX_CHECK(hipEventRecord(ev[0], st));
    if (stage_mask & 1) {
        if (int rc = per_group(1)) return rc;
        hipLaunchKernelGGL((k_ksh_finish<0>), dim3((1u << 14) / 256, a.nb * L), dim3(256), 0, st, a, a.nb * L);
    }
    if (ev) HX_CHECK(hipEventRecord(ev[1], st));
    if (stage_mask & 2) {
        a.nsel = L; a.selmap = 0xFEDCBA9876543210ull;
        if (int rc = launch_stage_h_tier<FUSED>(p, a, 2, tier_of(a.K - 1), p->x_skip)) return rc;
        hipLaunchKernelGGL((k_ksh_finish<1>), dim3((1u << 14) / 256, a.nb * 2), dim3(256), 0, st, a, a.nb * 2);
    }
    if (ev) HX_CHECK(hipEventRecord(ev[2], st));
    if (stage_mask & 4)
        if (int rc = per_group(4)) return rc;
    if (ev) HX_CHECK(hipEventRecord(ev[3], st));
    return (int)hipGetLastError();
}

size_t hx_ks_x_scratch_words(size_t L) { return L + 2; }   // per instance, in units of n (fits the (b, d)-major scratch)

// Large chunks of N = 16384 instances: one workgroup per (instance, limb) must fill the chip at least twice, like the
// fused k_ksf_up of the (b, d)-major pipeline. HEXL_KS_PIPE=1 keeps the (b, d)-major pipeline (tests, comparisons).
u32 hx_ks_x_loge() { return 4; }                                  // 16 coefficients per thread at every ring dimension
bool hx_ks_x_applies(const hexl_ks_plan* p, size_t nb) {
    static const int pipe = [] { const char* e = getenv("HEXL_KS_PIPE"); return e ? atoi(e) : 2; }();
    if (!p->d_keys_x || p->logn < 10 || p->logn > 15) return false;
    if (p->logn == 15 && p->x_loge != 4) return false;              // (N = 32768: the 16 x 1024 half-transform kernels only)
    // one workgroup per (instance, limb) must nearly fill the chip twice (a CU holds 16384 / N of them): measured at N = 16384,
    // L = 7 (tools/batch_sweep.py) the slot-major pipeline wins from 64 instances up (141 k against 132 k keyswitch/s), the
    // (b, d)-major one below 48
    return pipe == 3 || (pipe == 2 && ((4 * nb * p->L) << p->logn) >= ((7 * (size_t)p->ctx->num_cu) << 14));  // 3: always (tests)
}

// The plan's alias mask. Profiling builds (-DHEXL_PROFILING_AIDS: every stream, KX_ALIASED compiled into the kernels) read it from the
// environment here. Everywhere else it comes from hx_ksx_alias_mask() in alias_knob.hip -- a constant 0 in the shipped library. Aliasing
// the KEY rows needs no kernel support at all (key_stride is a launch argument), so libhexl_mi355x_keyalias.so links these very objects
// with a variant of that one function that honours bit 1: the key stream can then be taken out of the counters on kernels that are
// byte-identical to the shipped ones (round 5: the profiling build spills more and moves 19.2 MB per keyswitch where the shipped kernels
// move 14.4 -- differences taken ACROSS the two builds were wrong, profiles/r05_fetch_reconcile.json).
#ifdef HEXL_PROFILING_AIDS
static u32 ksx_alias_mask() {
    static const u32 mask = [] {
        const char* e = getenv("HEXL_KSX_ALIAS");
        const u32 m = e ? (u32)atoi(e) : 0u;
        if (m) fprintf(stderr, "[hexl_mi355x PROFILING BUILD] HEXL_KSX_ALIAS=%u: keyswitch results are WRONG by design\n", m);
        return m;
    }();
    return mask;
}
#else
u32 hx_ksx_alias_mask();
static u32 ksx_alias_mask() { return hx_ksx_alias_mask(); }
#endif

int hx_launch_keyswitch_x(hexl_ks_plan* p, u64* d_result, const u64* d_t_target, size_t nb, int stage_mask,
                          hipEvent_t* ev) {
    const size_t n = p->n, L = p->L;
    KsArgsX a;
    a.mods = p->d_mods_f64; a.tables = p->d_tables_f64; a.keys = p->d_keys_x;
    a.c = (double*)p->cur_scratch;
    a.s = a.c + p->cap * L * n;
    a.csub = a.s + p->cap * 2 * n;                                  // (N = 32768 only; inside the (b, d)-major scratch, which is larger)
    a.ssub = a.csub + p->cap * L * n;
    a.t_target = d_t_target; a.result = d_result;
    a.L = (u32)L; a.K = p->K; a.nb = (u32)nb;
    a.mul_a = a.mul_b = nullptr;
    a.stamps = nullptr;
    a.alias = ksx_alias_mask();
    a.key_stride = (a.alias & 1u) ? 0u : u32(2 * n);
    a.range_flag = p->d_flag;
    switch (p->logn) {
        case 10: return run_chunk_x<10, 4>(p, a, stage_mask, ev);
        case 11: return run_chunk_x<11, 4>(p, a, stage_mask, ev);
        case 12: return run_chunk_x<12, 4>(p, a, stage_mask, ev);
        case 13: return run_chunk_x<13, 4>(p, a, stage_mask, ev);
        case 14: break;
        case 15: return run_chunk_h(p, a, stage_mask, ev);           // two 16384-point halves per transform (k_ksh_*)
        default: return HEXL_E_BADARG;
    }
    // (the kernels' LAZY template argument = forward reduction period of the transforms, f64_arith.hpp: launch_stage_tier)
    return run_chunk_x<14, 4>(p, a, stage_mask, ev);
}

// fused ciphertext multiply + relinearize for one scratch chunk (always the slot-major pipeline, 16-coefficient geometry).
// Where the partial pass leaves a lane at most four adjacent words (N = 1024, 8192, 16384) the operand limbs are read straight at
// the transforms' B positions; N = 2048 / 4096 read them in A order and go through LDS (load_product_to_B, ksx_down_round).
// (N = 32768: the half-transform kernels k_ksh_*<..., FUSED>, hx_launch_mulrelin_x below.)
template <int LOGN>
static int mulrelin_for(hexl_ks_plan* p, const KsArgsX& a) { return run_chunk_x<LOGN, 4, true>(p, a, 7, nullptr); }
int hx_launch_mulrelin_x(hexl_ks_plan* p, u64* d_out, const u64* d_a, const u64* d_b, size_t nb) {
    const size_t n = p->n, L = p->L;
    if (!p->use_f64 || !p->d_keys_x || p->x_loge != 4) return HEXL_E_BADARG;
    KsArgsX a;
    a.mods = p->d_mods_f64; a.tables = p->d_tables_f64; a.keys = p->d_keys_x;
    a.c = (double*)p->cur_scratch;
    a.s = a.c + p->cap * L * n;
    a.csub = a.s + p->cap * 2 * n;                                  // (N = 32768 only)
    a.ssub = a.csub + p->cap * L * n;
    a.nsel = (u32)L; a.selmap = 0xFEDCBA9876543210ull;
    a.t_target = nullptr; a.result = d_out;
    a.L = (u32)L; a.K = p->K; a.nb = (u32)nb;
    a.mul_a = d_a; a.mul_b = d_b;
    a.stamps = nullptr;
    a.alias = 0;
    a.key_stride = u32(2 * n);
    a.range_flag = p->d_flag;
    switch (p->logn) {
        case 10: return mulrelin_for<10>(p, a);
        case 11: return mulrelin_for<11>(p, a);
        case 12: return mulrelin_for<12>(p, a);
        case 13: return mulrelin_for<13>(p, a);
        case 14: return mulrelin_for<14>(p, a);
        case 15: return run_chunk_h<true>(p, a, 7, nullptr);         // every transform as two 16384-point halves (k_ksh_*)
        default: return HEXL_E_BADARG;
    }
}
