// hexl_internal.hpp -- host-side state shared by the launcher translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "../../include/hexl_mi355x.h"
#include "f64_arith.hpp"

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

#define HX_CHECK(expr)                                                                       \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            fprintf(stderr, "[hexl_mi355x] %s failed: %s (%s:%d)\n", #expr,                  \
                    hipGetErrorString(_e), __FILE__, __LINE__);                              \
            return (int)_e;                                                                  \
        }                                                                                    \
    } while (0)

// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) applies to the CURRENT device's copy of a kernel, and the
// launchers may run on one host thread per device (NUM_DEV > 1): one of these per call site runs its initialiser
// once per device, serialised.
struct PerDeviceOnce {
    std::mutex m;
    uint64_t done = 0;
    template <class F>
    int run(int device, F init) {
        std::lock_guard<std::mutex> g(m);
        const uint64_t bit = 1ull << (device & 63);
        if (done & bit) return 0;
        const int rc = init();
        if (!rc) done |= bit;
        return rc;
    }
};

struct hexl_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // stream launches go to (own or caller's)
    int num_cu = 0;
    // grow-only device scratch + pinned staging used by the *_host entry points
    void* d_stage = nullptr;  size_t d_stage_bytes = 0;
    void* d_shared = nullptr; size_t d_shared_bytes = 0;   // small shared arrays of device-resident callers
    void* h_stage = nullptr;  size_t h_stage_bytes = 0;
    void* h_lone = nullptr;   size_t h_lone_bytes = 0;    // COHERENT pinned slabs of the zero-copy lone keyswitch (capi.hip keyswitch_host_lone)
    uint32_t lone_epoch = 0;                                // zero-copy lone keyswitches so far (their completion words carry it)
    // host-pointer pipeline: copy streams + events (created lazily), see run_pipeline() in capi.hip
    hipStream_t s_up = nullptr, s_down = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
    void* d_meta = nullptr;   size_t d_meta_bytes = 0;     // dyadic per-(item,modulus) constants
    void* d_ntt_redo = nullptr; size_t d_ntt_redo_bytes = 0; // N = 32768 standalone NTT: polynomials left to the integer butterflies (ntt.hip k_ntt_redo_*)
    void* d_ntt_tab = nullptr; size_t d_ntt_tab_bytes = 0;  // standalone NTT fast path: violation counters + derived double tables
    uint32_t ntt_seq = 0;                                   // launches so far (selects the violation counter)
    // "these tables are not Shoup tables" hints from the fast-path kernels to the host (ntt.hip NttHint): four pinned words
    unsigned long long *h_ntt_hint = nullptr, *d_ntt_hint = nullptr;
    unsigned long long* ntt_hint_word = nullptr; unsigned long long ntt_hint_tag = 0;   // of the launch being set up
    const uint32_t* ntt_clear_viol = nullptr;               // hinted route: the integer kernel clears the hint when this counter is 0
    char name[256] = {0};
};

int hx_reserve_device(hexl_ctx* ctx, void** p, size_t* cur, size_t need);
int hx_reserve_pinned(hexl_ctx* ctx, void** p, size_t* cur, size_t need, bool coherent = false);

// per-modulus constants of a keyswitch plan (device copy is an array of K of these)
struct KsModulus {
    u64 q;          // modulus
    u64 qbarr;      // floor(2^64/q)                      (fpga.cpp:1053)
    u64 inv_n, inv_n_p;       // n^-1 mod q and its Shoup factor (fpga.cpp:1070-1089)
    u64 inv_n_w, inv_n_w_p;   // n^-1 * W_last and its Shoup factor
    u64 msf, msf_p;           // modswitch factor reduced mod q (fpga.cpp:1057-1061) + Shoup factor
    u64 fix;        // q - (floor(q_sp/2) mod q)          (intt2_redu.hpp:31-32)
    u64 half;       // floor(q_sp/2)                      (intt2_redu.hpp:25)
    u64 len;        // floor(log2 q) - 1                  (128->64 Barrett, as fpga.cpp:366-373)
    u64 barr_lo;    // floor(2^(len+64)/q)
};

// the same constants for the FP64 path (keyswitch_f64.hip); residues centred, *_p = fl(value / p)
struct KsModF64 {
    hxf::Mod m;
    hxf::InvScale sc;
    double msf, msf_p;   // modswitch factor
    double fix;          // q - (floor(q_sp/2) mod q), in [1, q]
    double half;         // floor(q_sp/2)
};

// lanes (auxiliary streams) the chunks of one keyswitch call alternate between: 2 in production; HEXL_KS_LANES=3|4 is the
// experiment that bounds what a single launch per chunk could gain (tools/experiments/README.md, round 4)
constexpr int HX_KS_MAX_LANES = 4;
int hx_ks_lanes();

struct hexl_ks_plan {
    hexl_ctx* ctx = nullptr;
    u32 n = 0, logn = 0, L = 0, K = 0, rns = 0;
    std::vector<u64> moduli;
    KsModulus* d_mods = nullptr;      // [K]
    u64* d_tables = nullptr;          // [K][4][n]: roots, precon, inv_roots(HEXL idx), inv_precon
    u64* d_keys = nullptr;            // [L][L+1][2][n] in forward-output ("B") order
    u32 int_loge = 5;                 // elements-per-thread exponent of the integer kernels (fixes that B order)
    bool have_keys = false;
    // FP64 path (all moduli < 2^52): same tables / keys as centred doubles
    bool use_f64 = false;
    int f64_lazy = 0;                 // forward reduction period of the lazy kernels (3, 6, 12 by modulus size); 0 = strict -- the tier every
                                      // modulus of the plan admits (chosen from the LARGEST one)
    // Per-limb arithmetic tier (round 5): a transform runs modulo ONE q_i, so its reduction period depends on that modulus alone, as
    // every NTT engine of the reference runs on its own modulus (device/keyswitch/ntt_core.hpp:285-291, ntt1.hpp:107-128).
    // tier[i] = forward reduction period for limb i (0 = strict); `mixed` = some limb the plan uses admits a longer period than f64_lazy
    // (e.g. bridge-seal's chain 52,30,30,40,27,27,27: one strict limb, six at period 12). HEXL_KS_PER_LIMB=0 keeps the plan-wide tier.
    unsigned char tier[16] = {};
    bool mixed = false;
    u32 f64_loge = 4;                 // elements-per-thread exponent of the FP64 kernels (fixes the keys' B order)
    KsModF64* d_mods_f64 = nullptr;   // [K]
    double* d_tables_f64 = nullptr;   // [K][4][n]
    double* d_keys_f64 = nullptr;     // [L][L+1][2][n]
    double* d_keys_x = nullptr;       // the same keys in the B order of the slot-major pipeline's geometry (keyswitch_x.hip)
    u32 x_loge = 5;                   // ... whose elements-per-thread exponent this is
    // scratch for `cap` keyswitches per lane; two lanes (aux streams) work on alternating chunks so that kernels
    // of different kinds -- FP64-bound transforms and the HBM-bound multiply-accumulate -- share the chip and one
    // chunk's ragged last wave of workgroups is filled by the other's
    u64* d_scratch = nullptr;         // [lanes][cap * scratch_words * n]
    size_t cap = 0;
    hipStream_t aux[HX_KS_MAX_LANES] = {};
    hipEvent_t ev_start = nullptr, ev_done[HX_KS_MAX_LANES] = {};
    u32* d_flag = nullptr;            // two device words + their pinned host mirrors: [0] the kernels' input-range flag (HEXL_W_RANGE), [1] HEXL_KS_VALIDATE's (HEXL_E_RANGE)
    u32* h_flag = nullptr;
    double* d_keys_nat = nullptr;     // N = 16384 FP64 plans: the keys as centred doubles in NATURAL order (latency path, keyswitch_lat.hip)
    bool x_skip = false;              // slot-major lazy kernels: moduli within LAZY_SKIP_MAX_RATIO of each other -> c_d and s' enter the
                                      // transforms without a range reduction (keyswitch_x.hip SKIP variants; HEXL_KSX_SKIP=0 turns it off)
    bool overwrite_result = false;    // host-pointer path, (b, d)-major FP64 kernels: write `result` instead of accumulating into it
    // zero-copy lone keyswitch of the host-pointer entry point (capi.hip keyswitch_host_lone; set around ONE launch, else null):
    // per-quarter-limb completion words and the range flag in pinned host memory (keyswitch_lat.hip KsArgsQ)
    u32* host_done = nullptr;
    u32* host_flag = nullptr;
    u32 host_epoch = 0;
    hipStream_t cur = nullptr;        // stream the chunk being launched goes to
    u64* cur_scratch = nullptr;
};

// launcher prototypes implemented per translation unit
int hx_launch_ntt_fwd(hexl_ctx*, u64* d_x, size_t batch, const u64* roots, const u64* precon, u64 q, u64 n);
int hx_launch_ntt_inv(hexl_ctx*, u64* d_x, size_t batch, const u64* iroots, const u64* iprecon, u64 q,
                      u64 inv_n, u64 inv_n_p, u64 inv_n_w, u64 inv_n_w_p, u64 n);
int hx_launch_dyadic(hexl_ctx*, u64* d_out, const u64* d_a, const u64* d_b, size_t batch, u64 n,
                     const u64* d_moduli, u64 n_moduli);
int hx_launch_keyswitch(hexl_ks_plan*, u64* d_result, const u64* d_t_target, size_t batch, int stage_mask,
                        hipEvent_t* ev /* optional [4] */);
// one scratch chunk (nb <= plan->cap) on the FP64 path
int hx_launch_keyswitch_f64(hexl_ks_plan*, u64* d_result, const u64* d_t_target, size_t nb, int stage_mask,
                            hipEvent_t* ev);
// the same on the slot-major pipeline (keyswitch_x.hip; N = 16384, large chunks)
int hx_launch_keyswitch_x(hexl_ks_plan*, u64* d_result, const u64* d_t_target, size_t nb, int stage_mask,
                          hipEvent_t* ev);
bool hx_ks_x_applies(const hexl_ks_plan*, size_t nb);
size_t hx_ks_chunk(const hexl_ks_plan*);      // instances per scratch chunk (HEXL_KS_CHUNK or the default for the ring dimension)
// true when a batch of nb runs entirely on kernels that honour hexl_ks_plan::overwrite_result
bool hx_ks_can_overwrite(const hexl_ks_plan*, size_t nb);
// the lone-keyswitch latency path (keyswitch_lat.hip): N = 16384, FP64 plans; one instance per call on p->cur / p->cur_scratch
bool hx_ks_lat_applies(const hexl_ks_plan*, size_t nb);
int hx_launch_keyswitch_lat(hexl_ks_plan*, u64* d_result, const u64* d_t_target, size_t nb);
int hx_launch_multiply_relinearize(hexl_ks_plan*, u64* d_out, const u64* d_a, const u64* d_b, size_t batch);
u32 hx_ks_x_loge();
// index of coefficient held in register r of thread tid after a forward transform ("B layout")
u32 hx_idxB(u32 logn, u32 r, u32 tid);
u32 hx_loge_for(u32 logn);

static inline u64 hx_shoup(u64 y, u64 q) { return (u64)(((u128)(y % q) << 64) / q); }
