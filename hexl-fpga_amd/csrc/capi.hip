// capi.hip -- the extern "C" launcher ABI of include/hexl_mi355x.h: contexts, keyswitch plans
// (host precompute + upload), host-pointer staging variants, and the hipEvent timing hook.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>

#include "hexl_internal.hpp"
#include "number_theory.hpp"

// ------------------------------------------------------------------------------- context
int hx_reserve_device(hexl_ctx* ctx, void** p, size_t* cur, size_t need) {
    if (*cur >= need) return 0;
    if (*p) { HX_CHECK(hipStreamSynchronize(ctx->stream)); HX_CHECK(hipFree(*p)); *p = nullptr; *cur = 0; }
    HX_CHECK(hipMalloc(p, need));
    *cur = need;
    return 0;
}
int hx_reserve_pinned(hexl_ctx* ctx, void** p, size_t* cur, size_t need, bool coherent) {
    if (*cur >= need) return 0;
    if (*p) { HX_CHECK(hipStreamSynchronize(ctx->stream)); HX_CHECK(hipHostFree(*p)); *p = nullptr; *cur = 0; }
    // coherent (fine-grained) slabs are for the zero-copy lone keyswitch only, whose host side reads result limbs the RUNNING kernel has
    // just published (keyswitch_host_lone). The staging pipeline's slabs stay default: with coherent ones its copy-engine transfers ran a
    // third slower (worksize 128: 13.4 k against 21 k keyswitch/s through the C++ API)
    HX_CHECK(hipHostMalloc(p, need, coherent ? hipHostMallocCoherent : hipHostMallocDefault));
    *cur = need;
    return 0;
}

// NUMA node the library's own host threads (copy pool, unpack lane) ask for: the node of the devices in use when they all hang off one
// node, else no preference (host_simd.cpp; round 6: +16 % on the host-pointer KeySwitch at worksize 128). -2 = no context yet.
int hx_numa_node_of_pci(const char* bdf);
bool hx_pin_this_thread_to_node(int node);
static std::atomic<int> g_host_node{-2};
static void note_device_node(int device) {
    static const bool pin = [] { const char* e = getenv("HEXL_HOST_PIN"); return !(e && atoi(e) == 0); }();
    if (!pin) { g_host_node.store(-1); return; }
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); bdf[0] = 0; }
    for (char* c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = char(*c - 'A' + 'a');     // sysfs names are lower case
    const int node = hx_numa_node_of_pci(bdf);
    int seen = g_host_node.load();
    while (!g_host_node.compare_exchange_weak(seen, seen == -2 ? node : (seen == node ? node : -1))) {}
}
static void pin_own_thread() { (void)hx_pin_this_thread_to_node(g_host_node.load()); }

extern "C" int hexl_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count;
}

extern "C" int hexl_ctx_create(int device, hexl_ctx** out) {
    if (!out) return HEXL_E_BADARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) return HEXL_E_NODEVICE;
    HX_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HX_CHECK(hipGetDeviceProperties(&prop, device));
    hexl_ctx* c = new hexl_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    snprintf(c->name, sizeof(c->name), "hexl_mi355x: device %d %s (%s), %d CUs, %.1f GiB", device, prop.name,
             prop.gcnArchName, prop.multiProcessorCount, prop.totalGlobalMem / 1073741824.0);
    HX_CHECK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    note_device_node(device);
    *out = c;
    return 0;
}

extern "C" int hexl_ctx_destroy(hexl_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_shared) (void)hipFree(c->d_shared);
    if (c->d_meta) (void)hipFree(c->d_meta);
    if (c->d_ntt_tab) (void)hipFree(c->d_ntt_tab);
    if (c->d_ntt_redo) (void)hipFree(c->d_ntt_redo);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_lone) (void)hipHostFree(c->h_lone);
    if (c->h_ntt_hint) (void)hipHostFree(c->h_ntt_hint);
    if (c->s_up) (void)hipStreamDestroy(c->s_up);
    if (c->s_down) (void)hipStreamDestroy(c->s_down);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]);
        if (c->ev_comp[i]) (void)hipEventDestroy(c->ev_comp[i]);
        if (c->ev_down[i]) (void)hipEventDestroy(c->ev_down[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

extern "C" int hexl_ctx_set_stream(hexl_ctx* c, void* s) {
    if (!c) return HEXL_E_BADARG;
    c->stream = (hipStream_t)s;
    return 0;
}
extern "C" int hexl_ctx_use_own_stream(hexl_ctx* c) {
    if (!c) return HEXL_E_BADARG;
    c->stream = c->own_stream;
    return 0;
}
extern "C" int hexl_ctx_sync(hexl_ctx* c) {
    if (!c) return HEXL_E_BADARG;
    HX_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int hexl_ctx_describe(hexl_ctx* c, char* buf, size_t len) {
    if (!c || !buf || !len) return HEXL_E_BADARG;
    snprintf(buf, len, "%s", c->name);
    return 0;
}

// ------------------------------------------------------------------------------- K1..K3
// 1024..16384 as the reference (keyswitch) / 16384 (NTT); 32768 is beyond its envelope
static bool supported_ntt_n(u64 n) { return n == 1024 || n == 2048 || n == 4096 || n == 8192 || n == 16384 || n == 32768; }

extern "C" int hexl_ntt_fwd(hexl_ctx* c, uint64_t* x, size_t batch, const uint64_t* roots, const uint64_t* precon,
                            uint64_t q, uint64_t n) {
    if (!c || !x || !roots || !precon || !q || !supported_ntt_n(n) || batch > 0x7fffffffu) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(c->device));
    return hx_launch_ntt_fwd(c, x, batch, roots, precon, q, n);
}

extern "C" int hexl_ntt_inv(hexl_ctx* c, uint64_t* x, size_t batch, const uint64_t* ir, const uint64_t* ip, uint64_t q,
                            uint64_t inv_n, uint64_t inv_n_w, uint64_t n) {
    if (!c || !x || !ir || !ip || !q || !supported_ntt_n(n) || batch > 0x7fffffffu) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(c->device));
    // floor(y*2^64/q) once on the host; the reference recomputes it per element on the device
    // (MultiplyUIntModLazy3, device/mod_ops.hpp:135-151)
    return hx_launch_ntt_inv(c, x, batch, ir, ip, q, inv_n, hx_shoup(inv_n, q), inv_n_w, hx_shoup(inv_n_w, q), n);
}

extern "C" int hexl_dyadic_multiply(hexl_ctx* c, uint64_t* out, const uint64_t* a, const uint64_t* b, size_t batch,
                                    uint64_t n, const uint64_t* moduli, uint64_t n_moduli) {
    if (!c || !out || !a || !b || !moduli) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(c->device));
    return hx_launch_dyadic(c, out, a, b, batch, n, moduli, n_moduli);
}

// ------------------------------------------------------------------------------- K4 plan
// elements-per-thread exponent of the keyswitch kernels (must match run_chunk<> / run_chunk_f64<> dispatch)
static u32 ks_loge(u32 logn, bool f64) { return logn <= 10 ? 4 : (f64 && logn == 14 ? 4 : 5); }
static u32 ks_idxB(u32 logn, u32 loge, u32 r, u32 tid) {
    const u32 P = (logn + loge - 1) / loge, KL = logn - (P - 1) * loge;
    const u32 WB = logn - loge < 6 ? logn - loge : 6;          // mirrors Geom::idxB (ntt_core.hpp)
    const u32 grp = ((tid >> WB) << (loge - KL + WB)) + ((r >> KL) << WB) + (tid & ((1u << WB) - 1));
    return (grp << KL) + (r & ((1u << KL) - 1));
}
// perm[pos] = natural coefficient index stored at position pos of a "B order" ([r][tid]) limb
static std::vector<u32> ks_perm(u32 logn, u32 loge) {
    const u32 n = 1u << logn, E = 1u << loge, T = n >> loge;
    std::vector<u32> perm(n);
    for (u32 r = 0; r < E; ++r)
        for (u32 t = 0; t < T; ++t) perm[r * T + t] = ks_idxB(logn, loge, r, t);
    return perm;
}

extern "C" int hexl_ks_plan_create(hexl_ctx* c, uint64_t n, uint64_t L, uint64_t K, uint64_t rns, uint64_t kcc,
                                   const uint64_t* h_moduli, const uint64_t* h_modswitch, const uint64_t* h_twiddles,
                                   hexl_ks_plan** out) {
    if (!c || !out || !h_moduli || !h_modswitch) return HEXL_E_BADARG;
    if (!supported_ntt_n(n) || kcc != 2 || L == 0 || K < 2 || L >= K || K > 16 || rns == 0) return HEXL_E_BADARG;
    for (u64 i = 0; i < K; ++i)
        if (h_moduli[i] < (1ULL << 16) || h_moduli[i] >= (1ULL << 60)) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(c->device));
    u32 logn = 0;
    while ((1ULL << logn) < n) ++logn;

    hexl_ks_plan* p = new hexl_ks_plan();
    p->ctx = c; p->n = (u32)n; p->logn = logn; p->L = (u32)L; p->K = (u32)K; p->rns = (u32)rns;
    p->moduli.assign(h_moduli, h_moduli + K);

    const u64 q_sp = h_moduli[K - 1];
    std::vector<KsModulus> mods(K);
    std::vector<u64> tables(size_t(K) * 4 * n);
    std::vector<u64> inv0(n);
    for (u64 i = 0; i < K; ++i) {
        const u64 q = h_moduli[i];
        u64* roots = &tables[(i * 4 + 0) * n];
        u64* precon = &tables[(i * 4 + 1) * n];
        u64* iroots = &tables[(i * 4 + 2) * n];
        u64* iprecon = &tables[(i * 4 + 3) * n];
        if (h_twiddles) {
            // caller blocks [inv_roots | precon_inv | roots | precon_roots]; like the reference's
            // twiddle_generator.hpp:35-79 only blocks 0 and 2 are consumed
            memcpy(inv0.data(), h_twiddles + i * 4 * n, n * sizeof(u64));
            memcpy(roots, h_twiddles + (i * 4 + 2) * n, n * sizeof(u64));
        } else {
            const u64 w = hxnt::minimal_primitive_root(2 * n, q);      // fpga.cpp:1097-1109
            hxnt::forward_roots(n, logn, q, w, roots);
            hxnt::inverse_roots_from0(n, q, roots, inv0.data());
        }
        // internal inverse table uses the standalone kernel's indexing (first entry at index 1)
        iroots[0] = 1;
        for (u64 r = 1; r < n; ++r) iroots[r] = inv0[r - 1];
        for (u64 r = 0; r < n; ++r) {
            precon[r] = hx_shoup(roots[r], q);
            iprecon[r] = hx_shoup(iroots[r], q);
        }
        KsModulus& m = mods[i];
        m.q = q;
        m.qbarr = (u64)(((u128)1 << 64) / q);                           // fpga.cpp:1053
        m.inv_n = hxnt::invmod(n, q);                                   // fpga.cpp:1073
        m.inv_n_p = hx_shoup(m.inv_n, q);
        m.inv_n_w = hxnt::mulmod(m.inv_n, iroots[n - 1] % q, q);        // last-stage W folded into the scaling
        m.inv_n_w_p = hx_shoup(m.inv_n_w, q);
        u64 msf = h_modswitch[i];                                       // ReduceMod<8>, fpga.cpp:1057-1061
        if (msf >= 4 * q) msf -= 4 * q;
        if (msf >= 2 * q) msf -= 2 * q;
        if (msf >= q) msf -= q;
        m.msf = msf;
        m.msf_p = hx_shoup(msf, q);
        m.half = q_sp >> 1;
        m.fix = q - (m.half % q);                                       // intt2_redu.hpp:31-32
        u32 fl = 63;
        while (!(q >> fl)) --fl;
        m.len = fl - 1;
        m.barr_lo = (u64)(((u128)1 << (m.len + 64)) / q);
    }
    // FP64 path: every modulus below 2^52 (the reference's own bound); HEXL_KS_INT=1 forces the integer kernels
    bool f64_ok = !(getenv("HEXL_KS_INT") && atoi(getenv("HEXL_KS_INT")) == 1);
    for (u64 i = 0; i < K; ++i) f64_ok = f64_ok && h_moduli[i] < (1ULL << 52);
    if (logn == 15 && !f64_ok) { delete p; return HEXL_E_BADARG; }    // N = 32768: FP64 kernels only (moduli < 2^52)
    p->use_f64 = f64_ok;
    p->f64_lazy = 0;
    p->f64_loge = ks_loge(logn, true);
    p->int_loge = ks_loge(logn, false);
    if (logn == 14) {                                             // HEXL_KSI_LOGE=4: 16 coefficients x 1024 threads
        const char* e = getenv("HEXL_KSI_LOGE");
        if (e && (atoi(e) == 4 || atoi(e) == 5)) p->int_loge = (u32)atoi(e);
    }
    if (f64_ok && !(getenv("HEXL_KS_NOLAZY") && atoi(getenv("HEXL_KS_NOLAZY")) == 1)) {
        double qmax = 0;
        for (u64 i = 0; i < K; ++i) qmax = (double)h_moduli[i] > qmax ? (double)h_moduli[i] : qmax;
        p->f64_lazy = hxf::lazy_period_for(qmax);
        double qmin = qmax;
        for (u64 i = 0; i < K; ++i) qmin = (double)h_moduli[i] < qmin ? (double)h_moduli[i] : qmin;
        const char* sk = getenv("HEXL_KSX_SKIP");                   // 0: keep the range reductions (tests)
        p->x_skip = p->f64_lazy && qmax <= hxf::LAZY_SKIP_MAX_RATIO * qmin && !(sk && atoi(sk) == 0);
        const char* e = getenv("HEXL_KS_PERIOD");               // testing: force a SHORTER period (always valid)
        const int cap = (e && (atoi(e) == 3 || atoi(e) == 6 || atoi(e) == 12)) ? atoi(e) : 12;
        if (p->f64_lazy > cap) p->f64_lazy = cap;
        // per-limb tiers: limb i's transforms run modulo q_i alone (hexl_internal.hpp); HEXL_KS_PER_LIMB=0: the plan-wide tier for all
        const char* pl = getenv("HEXL_KS_PER_LIMB");
        const bool per_limb = !(pl && atoi(pl) == 0);
        for (u64 i = 0; i < K; ++i) {
            int t = per_limb ? hxf::lazy_period_for((double)h_moduli[i]) : p->f64_lazy;
            if (t > cap) t = cap;
            p->tier[i] = (unsigned char)t;
            if ((i < L || i == K - 1) && t != p->f64_lazy) p->mixed = true;
        }
    }
    if (f64_ok) {
        std::vector<KsModF64> fm(K);
        std::vector<double> ft(size_t(K) * 4 * n);
        for (u64 i = 0; i < K; ++i) {
            const u64 q = h_moduli[i];
            const double pd = (double)q;
            auto centre = [&](u64 v) { v %= q; return v > q / 2 ? (double)v - pd : (double)v; };
            for (int blk = 0; blk < 4; blk += 2)                   // blocks 0 / 2 hold w, blocks 1 / 3 hold fl(w/p)
                for (u64 r = 0; r < n; ++r) {
                    const double w = centre(tables[(i * 4 + blk) * n + r]);
                    ft[(i * 4 + blk) * n + r] = w;
                    ft[(i * 4 + blk + 1) * n + r] = w / pd;
                }
            KsModF64& f = fm[i];
            f.m.p = pd; f.m.pinv = 1.0 / pd;
            f.sc.n = centre(mods[i].inv_n);     f.sc.n_p = f.sc.n / pd;
            f.sc.nw = centre(mods[i].inv_n_w);  f.sc.nw_p = f.sc.nw / pd;
            f.msf = centre(mods[i].msf);        f.msf_p = f.msf / pd;
            f.fix = (double)mods[i].fix;
            f.half = (double)mods[i].half;
        }
        HX_CHECK(hipMalloc((void**)&p->d_mods_f64, K * sizeof(KsModF64)));
        HX_CHECK(hipMalloc((void**)&p->d_tables_f64, ft.size() * sizeof(double)));
        HX_CHECK(hipMalloc((void**)&p->d_keys_f64, size_t(L) * (L + 1) * 2 * n * sizeof(double)));
        if (logn >= 10 && logn <= 15) HX_CHECK(hipMalloc((void**)&p->d_keys_x, size_t(L) * (L + 1) * 2 * n * sizeof(double)));
        if (logn == 14) HX_CHECK(hipMalloc((void**)&p->d_keys_nat, size_t(L) * (L + 1) * 2 * n * sizeof(double)));
        HX_CHECK(hipMemcpy(p->d_mods_f64, fm.data(), K * sizeof(KsModF64), hipMemcpyHostToDevice));
        HX_CHECK(hipMemcpy(p->d_tables_f64, ft.data(), ft.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // input-range flag of the FP64 kernels (hexl_ks_range_check) + its pinned mirror; HEXL_KS_VALIDATE shares them
    // word 0: the kernels' range flag (HEXL_W_RANGE); word 1: HEXL_KS_VALIDATE's own (HEXL_E_RANGE) -- two statuses, two words
    // (word 2: the output gate of the zero-copy lone keyswitch, keyswitch_lat.hip k_ksq_down)
    HX_CHECK(hipMalloc((void**)&p->d_flag, 4 * sizeof(u32)));
    HX_CHECK(hipMemset(p->d_flag, 0, 4 * sizeof(u32)));
    HX_CHECK(hipHostMalloc((void**)&p->h_flag, 2 * sizeof(u32), hipHostMallocDefault));
    HX_CHECK(hipMalloc((void**)&p->d_mods, K * sizeof(KsModulus)));
    HX_CHECK(hipMalloc((void**)&p->d_tables, tables.size() * sizeof(u64)));
    // integer-kernel keys (key words + Shoup factors): only plans that run the integer kernels hold them
    if (!f64_ok) HX_CHECK(hipMalloc((void**)&p->d_keys, size_t(L) * (L + 1) * 4 * n * sizeof(u64)));
    HX_CHECK(hipMemcpy(p->d_mods, mods.data(), K * sizeof(KsModulus), hipMemcpyHostToDevice));
    HX_CHECK(hipMemcpy(p->d_tables, tables.data(), tables.size() * sizeof(u64), hipMemcpyHostToDevice));
    *out = p;
    return 0;
}

extern "C" int hexl_ks_plan_destroy(hexl_ks_plan* p) {
    if (!p) return 0;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    if (p->d_mods) (void)hipFree(p->d_mods);
    if (p->d_tables) (void)hipFree(p->d_tables);
    if (p->d_keys) (void)hipFree(p->d_keys);
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    for (int l = 0; l < HX_KS_MAX_LANES; ++l) {
        if (p->aux[l]) (void)hipStreamDestroy(p->aux[l]);
        if (p->ev_done[l]) (void)hipEventDestroy(p->ev_done[l]);
    }
    if (p->ev_start) (void)hipEventDestroy(p->ev_start);
    if (p->d_mods_f64) (void)hipFree(p->d_mods_f64);
    if (p->d_tables_f64) (void)hipFree(p->d_tables_f64);
    if (p->d_keys_f64) (void)hipFree(p->d_keys_f64);
    if (p->d_keys_x) (void)hipFree(p->d_keys_x);
    if (p->d_keys_nat) (void)hipFree(p->d_keys_nat);
    if (p->d_flag) (void)hipFree(p->d_flag);
    if (p->h_flag) (void)hipHostFree(p->h_flag);
    delete p;
    return 0;
}

// The reference packs keys into 3x256-bit DDR words (KeySwitch_load_keys, fpga.cpp:1167-1248); the GPU
// keeps plain 64-bit words but permutes each (d, slot, k) limb into the forward transform's register
// order so the key multiply-accumulate reads them fully coalesced.
extern "C" int hexl_ks_set_keys(hexl_ks_plan* p, const uint64_t* const* h_keys) {
    if (!p || !h_keys) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(p->ctx->device));
    const u64 n = p->n, L = p->L, K = p->K;
    // integer kernels: [d][slot][k][0] = key mod q, [..][1] = floor(that * 2^64 / q); FP64 plans never run them and hold
    // no integer copy (22 MB of device memory and a host pass per key set for nothing)
    std::vector<u64> dev;
    std::vector<u32> perm;
    const size_t words = size_t(L) * (L + 1) * 2 * n;
    if (!p->use_f64) { dev.resize(2 * words); perm = ks_perm(p->logn, p->int_loge); }
    std::vector<double> devf, devx, devn;
    if (p->d_keys_nat) devn.resize(words);
    std::vector<u32> permf, permx;
    if (p->use_f64) { devf.resize(words); permf = ks_perm(p->logn, p->f64_loge); }
    if (p->d_keys_x) {
        devx.resize(words);
        p->x_loge = p->logn == 14 ? hx_ks_x_loge() : 4;
        if (p->logn == 15) {
            // N = 32768: every transform of the slot-major pipeline is two 16384-point halves (keyswitch_x.hip k_ksh_*); NTT-domain block h
            // of a key row is kept in the B order of THAT geometry
            const std::vector<u32> half = ks_perm(14, 4);
            permx.resize(n);
            for (u32 h = 0; h < 2; ++h)
                for (u32 j = 0; j < (1u << 14); ++j) permx[h * (1u << 14) + j] = h * (1u << 14) + half[j];
        } else {
            permx = ks_perm(p->logn, p->x_loge);
        }
    }
    for (u64 d = 0; d < L; ++d) {
        if (!h_keys[d]) return HEXL_E_BADARG;
        for (u64 slot = 0; slot <= L; ++slot) {
            const u64 i = slot < L ? slot : K - 1;
            const u64 q = p->moduli[i];
            for (u64 k = 0; k < 2; ++k) {
                const u64* src = h_keys[d] + (k * K + i) * n;            // fpga.cpp:1186-1190
                const size_t base = ((d * (L + 1) + slot) * 2 + k) * n;
                if (!p->use_f64)
                    for (u64 j = 0; j < n; ++j) {
                        const u64 v = src[perm[j]] % q;
                        dev[2 * base + j] = v;
                        dev[2 * base + n + j] = (u64)(((u128)v << 64) / q);
                    }
                if (p->use_f64)                                          // same limb as centred doubles
                    for (u64 j = 0; j < n; ++j) {
                        const u64 v = src[permf[j]] % q;
                        devf[base + j] = v > q / 2 ? (double)v - (double)q : (double)v;
                    }
                if (p->d_keys_x)
                    for (u64 j = 0; j < n; ++j) {
                        const u64 v = src[permx[j]] % q;
                        devx[base + j] = v > q / 2 ? (double)v - (double)q : (double)v;
                    }
                if (p->d_keys_nat)                                       // natural order (latency path)
                    for (u64 j = 0; j < n; ++j) {
                        const u64 v = src[j] % q;
                        devn[base + j] = v > q / 2 ? (double)v - (double)q : (double)v;
                    }
            }
        }
    }
    HX_CHECK(hipStreamSynchronize(p->ctx->stream));
    if (!p->use_f64) HX_CHECK(hipMemcpy(p->d_keys, dev.data(), dev.size() * sizeof(u64), hipMemcpyHostToDevice));
    if (p->use_f64)
        HX_CHECK(hipMemcpy(p->d_keys_f64, devf.data(), devf.size() * sizeof(double), hipMemcpyHostToDevice));
    if (p->d_keys_x)
        HX_CHECK(hipMemcpy(p->d_keys_x, devx.data(), devx.size() * sizeof(double), hipMemcpyHostToDevice));
    if (p->d_keys_nat)
        HX_CHECK(hipMemcpy(p->d_keys_nat, devn.data(), devn.size() * sizeof(double), hipMemcpyHostToDevice));
    p->have_keys = true;
    return 0;
}

extern "C" int hexl_keyswitch(hexl_ks_plan* p, uint64_t* d_result, const uint64_t* d_t_target, size_t batch) {
    if (!p || !d_result || !d_t_target) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(p->ctx->device));
    return hx_launch_keyswitch(p, d_result, d_t_target, batch, 7, nullptr);
}

extern "C" int hexl_ks_plan_tiers(const hexl_ks_plan* p, int* tiers) {
    if (!p || !tiers) return HEXL_E_BADARG;
    for (u32 i = 0; i < p->K; ++i) tiers[i] = p->use_f64 ? (int)p->tier[i] : -1;
    return p->use_f64 && p->mixed ? 1 : 0;
}

extern "C" int hexl_ks_range_check(hexl_ks_plan* p) {
    if (!p) return HEXL_E_BADARG;
    hexl_ctx* c = p->ctx;
    HX_CHECK(hipSetDevice(c->device));
    HX_CHECK(hipMemcpyAsync(p->h_flag, p->d_flag, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HX_CHECK(hipMemsetAsync(p->d_flag, 0, sizeof(u32), c->stream));
    HX_CHECK(hipStreamSynchronize(c->stream));
    return *p->h_flag ? HEXL_W_RANGE : 0;
}

extern "C" int hexl_multiply_relinearize(hexl_ks_plan* p, uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b,
                                         size_t batch) {
    if (!p || !d_out || !d_a || !d_b) return HEXL_E_BADARG;
    // d_out is written while d_a / d_b are still being read (component 0 is stored before component 1's operands are loaded)
    const size_t bytes = batch * 2 * size_t(p->L) * p->n * sizeof(u64);
    auto overlaps = [&](const uint64_t* x) {
        return (const char*)d_out < (const char*)x + bytes && (const char*)x < (const char*)d_out + bytes;
    };
    if (overlaps(d_a) || overlaps(d_b)) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(p->ctx->device));
    return hx_launch_multiply_relinearize(p, d_out, d_a, d_b, batch);
}

extern "C" int hexl_ks_time_stages(hexl_ks_plan* p, uint64_t* d_result, const uint64_t* d_t_target, size_t batch,
                                   int iters, float* ms_out) {
    if (!p || !ms_out || iters <= 0) return HEXL_E_BADARG;
    HX_CHECK(hipSetDevice(p->ctx->device));
    hipEvent_t ev[4];
    for (auto& e : ev) HX_CHECK(hipEventCreate(&e));
    float acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        int rc = hx_launch_keyswitch(p, d_result, d_t_target, batch, 7, ev);
        if (rc) return rc;
        HX_CHECK(hipEventSynchronize(ev[3]));
        float t;
        HX_CHECK(hipEventElapsedTime(&t, ev[0], ev[3])); acc[0] += t;
        for (int s = 0; s < 3; ++s) { HX_CHECK(hipEventElapsedTime(&t, ev[s], ev[s + 1])); acc[1 + s] += t; }
    }
    for (int s = 0; s < 4; ++s) ms_out[s] = acc[s] / iters;
    for (auto& e : ev) (void)hipEventDestroy(e);
    return 0;
}

// ------------------------------------------------------------------------------- host-pointer variants
// The reference stages every batch through device-visible memory (FPGAObject_*::fill_in_data / fill_out_data,
// host/src/fpga.cpp:329-518) with one batch in flight while the previous one is read back (:1517-1545). Here a
// batch is cut into sub-batches that flow through a 3-stage pipeline with double-buffered pinned and device
// slabs: CPU pack (parallel memcpy) -> H2D on its own stream -> kernels on the context stream -> D2H on a third
// stream -> CPU unpack, so PCIe traffic in both directions, the kernels and the host copies overlap.
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <future>
#include <initializer_list>
#include <utility>
#include <vector>
#include <deque>
#include <mutex>
#include <condition_variable>
#include <memory>

// the host's `result += output` per limb (host_simd.cpp)
void hx_add_mod_u64(uint64_t* r, const uint64_t* o, size_t n, uint64_t q);

static unsigned host_threads() {                                   // (thread-safe static: several device runners call this)
    static const unsigned n = [] {
        const char* e = getenv("HEXL_HOST_THREADS");
        // 8: swept on the 256-thread boxes of this pool (round 3, HEXL_HOST_THREADS sweep of tests/cpp/bench_cxx_api): 2 / 4 / 8 / 12 / 32 / 64 copy threads move 12 / 21 / 26 / 27 /
        // 19 / 12 k keyswitch/s through the host-pointer API at worksize 1024 -- the copies are memory-bound, more threads only contend
        // (round 4: those pods run under a CFS quota of 16 cores, bench.py cpu_baseline.cgroup_cpu_quota_cores -- a host without one may want more)
        const unsigned v = e ? (unsigned)atoi(e) : std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
        return v ? v : 1u;
    }();
    return n;
}

// Persistent host workers for the staging copies (spawning threads per sub-batch cost more than the copies:
// 64 threads measured slower than 8). Jobs are index ranges; several jobs may be in flight (the unpack of one
// sub-batch runs beside the pack of the next), workers drain them front to back.
namespace {
struct HostJob {
    std::function<void(size_t)> fn;
    size_t count = 0;
    std::atomic<size_t> next{0}, done{0};
};
class HostPool {
  public:
    static HostPool& get() { static HostPool p; return p; }
    std::shared_ptr<HostJob> submit(size_t count, std::function<void(size_t)> fn) {
        auto j = std::make_shared<HostJob>();
        j->fn = std::move(fn); j->count = count;
        if (!count) return j;
        { std::lock_guard<std::mutex> g(m_); jobs_.push_back(j); }
        cv_.notify_all();
        return j;
    }
    void wait(const std::shared_ptr<HostJob>& j) {
        if (!j) return;
        work_on(j.get());                                         // the caller helps instead of sleeping
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [&] { return j->done.load() >= j->count; });
    }
  private:
    HostPool() {
        const unsigned n = host_threads();
        for (unsigned t = 0; t + 1 < n; ++t) th_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void work_on(HostJob* j) {
        for (size_t i; (i = j->next.fetch_add(1)) < j->count;) {
            j->fn(i);
            if (j->done.fetch_add(1) + 1 == j->count) { std::lock_guard<std::mutex> g(m_); done_cv_.notify_all(); }
        }
    }
    void loop() {
        pin_own_thread();                                             // (the pool is created by the first copy: contexts exist by then)
        for (;;) {
            std::shared_ptr<HostJob> j;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] {
                    while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->count) jobs_.pop_front();
                    return stop_ || !jobs_.empty();
                });
                if (stop_) return;
                j = jobs_.front();
            }
            work_on(j.get());
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::deque<std::shared_ptr<HostJob>> jobs_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};
}  // namespace

static void parallel_for(size_t count, const std::function<void(size_t)>& fn) {
    if (count <= 1 || host_threads() <= 1) { for (size_t i = 0; i < count; ++i) fn(i); return; }
    HostPool::get().wait(HostPool::get().submit(count, fn));
}

// One persistent helper thread per calling thread (= per device runner): runs the posted jobs in order. The unpack of
// sub-batch k runs here beside the pack of sub-batch k + 2; a thread spawned per sub-batch (std::async) cost ~40 us each,
// a third of a single keyswitch's end-to-end time at worksize 1.
namespace {
class AsyncLane {
  public:
    AsyncLane() : th_([this] { loop(); }) {}
    ~AsyncLane() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    size_t post(std::function<void()> fn) {                       // returns the job's ticket
        size_t t;
        { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(fn)); t = ++posted_; }
        cv_.notify_all();
        return t;
    }
    void wait(size_t ticket) {
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [&] { return finished_ >= ticket; });
    }
  private:
    void loop() {
        pin_own_thread();
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [&] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;                               // stop requested and nothing left
            std::function<void()> fn = std::move(q_.front());
            q_.pop_front();
            g.unlock();
            fn();
            g.lock();
            ++finished_;
            done_cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::deque<std::function<void()>> q_;
    size_t posted_ = 0, finished_ = 0;
    bool stop_ = false;
    std::thread th_;                                              // last member: starts after the others exist
};
}  // namespace

static int pipe_init(hexl_ctx* c) {
    if (c->s_up) return 0;
    HX_CHECK(hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking));
    HX_CHECK(hipStreamCreateWithFlags(&c->s_down, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HX_CHECK(hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
        HX_CHECK(hipEventCreateWithFlags(&c->ev_comp[i], hipEventDisableTiming));
        HX_CHECK(hipEventCreateWithFlags(&c->ev_down[i], hipEventDisableTiming));
    }
    return 0;
}

// pack(first, count, h_in)      fill the pinned input slab for items [first, first+count)
// compute(count, d_in, d_out)   enqueue kernels on c->stream
// unpack(first, count, h_out)   consume the pinned output slab
// `shared_bytes` at the head of every input slab is filled by pack_shared once per slab (tables, moduli).
struct PipeShape { size_t in1, out1, shared, sub; bool in_place; };

static int run_pipeline(hexl_ctx* c, size_t batch, const PipeShape& sh,
                        const std::function<void(char*)>& pack_shared,
                        const std::function<void(size_t, size_t, char*)>& pack,
                        const std::function<int(size_t, char*, char*)>& compute,
                        const std::function<void(size_t, size_t, const char*)>& unpack) {
    int rc = pipe_init(c);
    if (rc) return rc;
    // HEXL_HOST_TRACE=1: microseconds since the call started at each step of the staging pipeline (stderr)
    static const bool trace = [] { const char* e = getenv("HEXL_HOST_TRACE"); return e && atoi(e) == 1; }();
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what, size_t k) {
        if (trace)
            fprintf(stderr, "[hexl host] %-22s sub-batch %zu  +%8.1f us  @%12.1f us\n", what, k,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(),
                    fmod(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(), 1e8));
    };
    // at least four sub-batches once there are four items -- eight from 64 items up -- so that the stages overlap for small windows
    // too: the window's first upload and last unpack are exposed, and they shrink with the sub-batch (round 6 trace at worksize 128,
    // profiles/r06_host_trace_worksize128.txt: 1.7 ms of ramp-up with four sub-batches of 32)
    const size_t S = std::min(sh.sub, std::max<size_t>(1, batch >= 64 ? (batch + 7) / 8 : (batch + 3) / 4));
    const size_t in_slab = (sh.shared + S * sh.in1 + 255) & ~size_t(255);
    // in-place primitives compute inside the device input slab, but on the HOST side the download areas are separate from the upload
    // areas and there are FOUR of them (two of everything else): the unpack -- for a keyswitch the host's `result +=`, the slowest
    // stage of the pipeline -- of sub-batch k then only has to be finished before the download of sub-batch k + 4 is enqueued, not k + 2
    // (round 6: with two, every other download waited for an unpack and the copy engine idled behind the host)
    constexpr size_t HOUT = 4;
    const size_t out_slab = (S * (sh.in_place ? sh.in1 : sh.out1) + 255) & ~size_t(255);
    const size_t dset = in_slab + (sh.in_place ? 0 : out_slab);
    rc = hx_reserve_device(c, &c->d_stage, &c->d_stage_bytes, 2 * dset);
    if (!rc) rc = hx_reserve_pinned(c, &c->h_stage, &c->h_stage_bytes, 2 * in_slab + HOUT * out_slab);
    if (rc) return rc;
    // Sub-batch sizes: S each, but from 64 objects up the window OPENS with S/4 and S/2 and CLOSES with S/2 and S/4: the first upload +
    // kernels + download and the last unpack are exposed (nothing to overlap with), and they shrink with their sub-batch -- in steady
    // state the pipeline runs at the PCIe download rate (round 6: 16 objects per 0.55 ms at L = 6 = 45 GB/s), so a window costs that
    // plus ramp and tail (worksize 128: 1.2 ms of ramp with eight equal sub-batches)
    std::vector<size_t> sub_first, sub_cnt;
    {
        std::vector<size_t> head, tail;
        if (batch >= 64 && S >= 8) { head = {S / 4, S / 2}; tail = {S / 2, S / 4}; }
        size_t at = 0, closing = 0;
        for (size_t t : tail) closing += t;
        for (size_t h : head) { sub_first.push_back(at); sub_cnt.push_back(h); at += h; }
        while (at + closing < batch) { const size_t cnt = std::min(S, batch - closing - at); sub_first.push_back(at); sub_cnt.push_back(cnt); at += cnt; }
        for (size_t t : tail) { sub_first.push_back(at); sub_cnt.push_back(t); at += t; }
    }
    const size_t nsub = sub_first.size();
    auto h_in = [&](size_t k) { return (char*)c->h_stage + (k & 1) * in_slab; };
    auto d_in = [&](size_t k) { return (char*)c->d_stage + (k & 1) * dset; };
    auto h_out = [&](size_t k) { return (char*)c->h_stage + 2 * in_slab + (k % HOUT) * out_slab; };
    auto d_out = [&](size_t k) { return sh.in_place ? d_in(k) + sh.shared : d_in(k) + in_slab; };
    if (nsub == 1) {
        // one sub-batch: nothing to overlap with -- one stream, one synchronisation, the unpack on the calling thread
        if (sh.shared) pack_shared(h_in(0));
        pack(0, batch, h_in(0) + sh.shared);
        stamp("packed", 0);
        HX_CHECK(hipMemcpyAsync(d_in(0), h_in(0), sh.shared + batch * sh.in1, hipMemcpyHostToDevice, c->stream));
        rc = compute(batch, d_in(0), d_out(0));
        if (rc) return rc;
        HX_CHECK(hipMemcpyAsync(h_out(0), d_out(0), batch * (sh.in_place ? sh.in1 : sh.out1), hipMemcpyDeviceToHost, c->stream));
        stamp("enqueued", 0);
        HX_CHECK(hipStreamSynchronize(c->stream));
        stamp("download complete", 0);
        unpack(0, batch, h_out(0));
        stamp("done", 1);
        return 0;
    }
    // unpack(k) runs on the helper lane beside pack(k+2), strictly in submission order (a keyswitch unpack ACCUMULATES
    // into the caller's result, and the same result may appear in several sub-batches, benchmark/bench_keyswitch.cpp:
    // 113-131); it must be finished before the download of sub-batch k + HOUT is enqueued (same host download area)
    static thread_local AsyncLane lane;
    size_t ticket[HOUT] = {}, last_ticket = 0;
    struct Drain {                                                // the posted jobs reference this frame: never leave before them
        AsyncLane& l; size_t& t;
        ~Drain() { l.wait(t); }
    } drain{lane, last_ticket};
    for (size_t it = 0; it < nsub + 2; ++it) {
        if (it >= 2) {                                            // drain sub-batch it-2 (frees slab set it&1)
            const size_t k = it - 2, first = sub_first[k], cnt = sub_cnt[k];
            HX_CHECK(hipEventSynchronize(c->ev_down[k & 1]));
            stamp("download complete", k);
            const char* src = h_out(k);
            last_ticket = ticket[k % HOUT] = lane.post([&unpack, &stamp, first, cnt, src, k] { unpack(first, cnt, src); stamp("unpacked", k); });
        }
        if (it < nsub) {
            const size_t first = sub_first[it], cnt = sub_cnt[it];
            if (sh.shared) pack_shared(h_in(it));
            pack(first, cnt, h_in(it) + sh.shared);
            stamp("packed", it);
            HX_CHECK(hipMemcpyAsync(d_in(it), h_in(it), sh.shared + cnt * sh.in1, hipMemcpyHostToDevice, c->s_up));
            HX_CHECK(hipEventRecord(c->ev_up[it & 1], c->s_up));
            HX_CHECK(hipStreamWaitEvent(c->stream, c->ev_up[it & 1], 0));
            rc = compute(cnt, d_in(it), d_out(it));
            if (rc) break;
            HX_CHECK(hipEventRecord(c->ev_comp[it & 1], c->stream));
            HX_CHECK(hipStreamWaitEvent(c->s_down, c->ev_comp[it & 1], 0));
            lane.wait(ticket[it % HOUT]);                         // (the unpack of sub-batch it - HOUT: this download area's last reader)
            HX_CHECK(hipMemcpyAsync(h_out(it), d_out(it), cnt * (sh.in_place ? sh.in1 : sh.out1), hipMemcpyDeviceToHost,
                                    c->s_down));
            HX_CHECK(hipEventRecord(c->ev_down[it & 1], c->s_down));
            stamp("enqueued", it);
        }
    }
    lane.wait(last_ticket);
    stamp("done", nsub);
    return rc;
}

static size_t sub_batch_for(size_t bytes_per_item) {              // ~128 MB slabs (32 MB measured 15-25 % slower at worksize 1024)
    const char* e = getenv("HEXL_HOST_SUB_MB");
    const size_t target = (e ? (size_t)atoi(e) : 128) << 20;
    return std::max<size_t>(1, target / std::max<size_t>(1, bytes_per_item));
}

// ------------------------------------------------------------------------------- device-resident callers
// The reference's API takes plain pointers; callers that already keep their ciphertexts in device (or managed)
// memory can pass those pointers to the same entry points: no staging, the kernels run where the data lies.
// All payload pointers of one call must be of the same kind; small shared arrays (tables, moduli) may be either.
static bool on_device(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain host memory
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}
// 0: every pointer is host memory, 1: every pointer is device memory, -1: mixed
static int payload_kind(std::initializer_list<std::pair<const void* const*, size_t>> lists) {
    size_t dev = 0, total = 0;
    for (auto& l : lists)
        for (size_t i = 0; i < l.second; ++i) { dev += on_device(l.first[i]) ? 1 : 0; ++total; }
    return dev == 0 ? 0 : dev == total ? 1 : -1;
}
// a device copy of a small shared array that may live on either side (`slot` = which quarter of the scratch)
static int shared_to_device(hexl_ctx* c, const void* src, size_t bytes, int slot, const void** out) {
    if (on_device(src)) { *out = src; return 0; }
    const size_t quarter = (bytes + 255) & ~size_t(255);
    int rc = hx_reserve_device(c, &c->d_shared, &c->d_shared_bytes, 4 * quarter);
    if (rc) return rc;
    char* dst = (char*)c->d_shared + slot * quarter;
    HX_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    *out = dst;
    return 0;
}

extern "C" int hexl_ntt_fwd_host(hexl_ctx* c, uint64_t* const* h_x, size_t batch, const uint64_t* h_roots,
                                 const uint64_t* h_precon, uint64_t q, uint64_t n) {
    if (!c || !h_x || !h_roots || !h_precon || !supported_ntt_n(n)) return HEXL_E_BADARG;
    if (!batch) return 0;
    HX_CHECK(hipSetDevice(c->device));
    const size_t one = n * 8;
    const int kind = payload_kind({{(const void* const*)h_x, batch}});
    if (kind < 0) return HEXL_E_BADARG;
    if (kind == 1) {                                               // device-resident polynomials: maximal contiguous runs
        const void *dr, *dp;
        int rc = shared_to_device(c, h_roots, one, 0, &dr);
        if (!rc) rc = shared_to_device(c, h_precon, one, 1, &dp);
        for (size_t i = 0, j; !rc && i < batch; i = j) {
            for (j = i + 1; j < batch && h_x[j] == h_x[j - 1] + n; ++j) {}
            rc = hexl_ntt_fwd(c, h_x[i], j - i, (const u64*)dr, (const u64*)dp, q, n);
        }
        return rc ? rc : (int)hipStreamSynchronize(c->stream);
    }
    PipeShape sh{one, one, 2 * one, sub_batch_for(one), true};
    return run_pipeline(c, batch, sh,
        [&](char* h) { memcpy(h, h_roots, one); memcpy(h + one, h_precon, one); },
        [&](size_t first, size_t cnt, char* h) { parallel_for(cnt, [&](size_t b) { memcpy(h + b * one, h_x[first + b], one); }); },
        [&](size_t cnt, char* d, char* dout) { return hexl_ntt_fwd(c, (u64*)dout, cnt, (u64*)d, (u64*)(d + one), q, n); },
        [&](size_t first, size_t cnt, const char* h) { parallel_for(cnt, [&](size_t b) { memcpy(h_x[first + b], h + b * one, one); }); });
}

extern "C" int hexl_ntt_inv_host(hexl_ctx* c, uint64_t* const* h_x, size_t batch, const uint64_t* h_ir,
                                 const uint64_t* h_ip, uint64_t q, uint64_t inv_n, uint64_t inv_n_w, uint64_t n) {
    if (!c || !h_x || !h_ir || !h_ip || !supported_ntt_n(n)) return HEXL_E_BADARG;
    if (!batch) return 0;
    HX_CHECK(hipSetDevice(c->device));
    const size_t one = n * 8;
    const int kind = payload_kind({{(const void* const*)h_x, batch}});
    if (kind < 0) return HEXL_E_BADARG;
    if (kind == 1) {
        const void *dr, *dp;
        int rc = shared_to_device(c, h_ir, one, 0, &dr);
        if (!rc) rc = shared_to_device(c, h_ip, one, 1, &dp);
        for (size_t i = 0, j; !rc && i < batch; i = j) {
            for (j = i + 1; j < batch && h_x[j] == h_x[j - 1] + n; ++j) {}
            rc = hexl_ntt_inv(c, h_x[i], j - i, (const u64*)dr, (const u64*)dp, q, inv_n, inv_n_w, n);
        }
        return rc ? rc : (int)hipStreamSynchronize(c->stream);
    }
    PipeShape sh{one, one, 2 * one, sub_batch_for(one), true};
    return run_pipeline(c, batch, sh,
        [&](char* h) { memcpy(h, h_ir, one); memcpy(h + one, h_ip, one); },
        [&](size_t first, size_t cnt, char* h) { parallel_for(cnt, [&](size_t b) { memcpy(h + b * one, h_x[first + b], one); }); },
        [&](size_t cnt, char* d, char* dout) { return hexl_ntt_inv(c, (u64*)dout, cnt, (u64*)d, (u64*)(d + one), q, inv_n, inv_n_w, n); },
        [&](size_t first, size_t cnt, const char* h) { parallel_for(cnt, [&](size_t b) { memcpy(h_x[first + b], h + b * one, one); }); });
}

extern "C" int hexl_dyadic_multiply_host(hexl_ctx* c, uint64_t* const* h_out, const uint64_t* const* h_a,
                                         const uint64_t* const* h_b, size_t batch, uint64_t n,
                                         const uint64_t* const* h_moduli, uint64_t n_moduli) {
    if (!c || !h_out || !h_a || !h_b || !h_moduli) return HEXL_E_BADARG;
    if (!batch) return 0;
    HX_CHECK(hipSetDevice(c->device));
    const size_t op1 = 2 * n_moduli * n * 8, out1 = 3 * n_moduli * n * 8, mod1 = n_moduli * 8;
    const int kind = payload_kind({{(const void* const*)h_out, batch}, {(const void* const*)h_a, batch},
                                   {(const void* const*)h_b, batch}});
    if (kind < 0) return HEXL_E_BADARG;
    if (kind == 1) {                                               // device-resident operands and results
        int rc = 0;
        for (size_t i = 0, j; !rc && i < batch; i = j) {
            for (j = i + 1; j < batch && h_out[j] == h_out[j - 1] + out1 / 8 && h_a[j] == h_a[j - 1] + op1 / 8 &&
                            h_b[j] == h_b[j - 1] + op1 / 8; ++j) {}
            // per-item moduli: gather the run's lists into one device array
            std::vector<u64> mods((j - i) * n_moduli);
            bool dev_mod = on_device(h_moduli[i]);
            const void* dm = nullptr;
            if (dev_mod) {                                         // must then be one contiguous device array, too
                for (size_t k = i + 1; k < j; ++k) dev_mod = dev_mod && h_moduli[k] == h_moduli[k - 1] + n_moduli;
                if (!dev_mod) return HEXL_E_BADARG;
                dm = h_moduli[i];
            } else {
                for (size_t k = i; k < j; ++k) memcpy(&mods[(k - i) * n_moduli], h_moduli[k], mod1);
                rc = hx_reserve_device(c, &c->d_shared, &c->d_shared_bytes, mods.size() * 8);
                if (rc) return rc;
                HX_CHECK(hipMemcpy(c->d_shared, mods.data(), mods.size() * 8, hipMemcpyHostToDevice));
                dm = c->d_shared;
            }
            rc = hexl_dyadic_multiply(c, h_out[i], h_a[i], h_b[i], j - i, n, (const u64*)dm, n_moduli);
            if (!rc) rc = (int)hipStreamSynchronize(c->stream);    // d_shared is reused by the next run
        }
        return rc;
    }
    const size_t in1 = 2 * op1 + ((mod1 + 15) & ~size_t(15));   // per item: [a | b | moduli(padded)]
    PipeShape sh{in1, out1, 0, sub_batch_for(in1 + out1), false};
    // layout inside a slab holding cnt items: a[cnt] | b[cnt] | moduli[cnt] (contiguous batches for the kernel);
    // cnt*(2*op1 + mod1) <= cnt*in1 bytes, which is what the pipeline uploads
    return run_pipeline(c, batch, sh, [](char*) {},
        [&](size_t first, size_t cnt, char* h) {
            parallel_for(cnt, [&](size_t k) {
                memcpy(h + k * op1, h_a[first + k], op1);
                memcpy(h + cnt * op1 + k * op1, h_b[first + k], op1);
                memcpy(h + 2 * cnt * op1 + k * mod1, h_moduli[first + k], mod1);
            });
        },
        [&](size_t cnt, char* d, char* dout) {
            return hexl_dyadic_multiply(c, (u64*)dout, (u64*)d, (u64*)(d + cnt * op1), cnt, n, (u64*)(d + 2 * cnt * op1),
                                        n_moduli);
        },
        [&](size_t first, size_t cnt, const char* h) { parallel_for(cnt, [&](size_t k) { memcpy(h_out[first + k], h + k * out1, out1); }); });
}

// A LONE keyswitch (the SEAL bridge's call shape: set_worksize_KeySwitch(1), experimental/bridge-seal/tests/fpga_context.h:13-16) through
// the host-pointer entry point, ZERO-COPY (round 5). The staged pipeline spent 212 us per call at L = 6: pack 10, seven enqueues 22,
// [hipMemcpyAsync 0.8 MB up 29.5, four kernels 50, hipMemcpyAsync 1.6 MB down 43, gaps] 121, host accumulate 37, hand-overs 20
// (HEXL_HOST_TRACE=1). Here the kernels of the quarter-transform path (keyswitch_lat.hip) work on the pinned slabs themselves:
// k_ksq_intt pulls t_target across PCIe while it transforms, k_ksq_down pushes the output the same way (tools/zero_copy_probe: 18.7 us
// and 31.6 us for those bytes, overlapped with the arithmetic, against 29.5 + 43.2 us of copies around it) and publishes every finished
// quarter limb in a pinned word, so the host adds limb i into the caller's `result` (the reference's contract: the HOST accumulates,
// FPGAObject_KeySwitch::fill_out_data, fpga.cpp:441-475) while limbs i + 1 ... are still in flight. No copy engine, no memset, no
// stream synchronisation on the critical path; the range flag is a pinned word too.
static int keyswitch_host_lone(hexl_ks_plan* p, uint64_t* const* h_results, const uint64_t* const* h_t_targets, size_t batch) {
    hexl_ctx* c = p->ctx;
    const size_t n = p->n, L = p->L, tt = L * n * 8, rs = 2 * tt;
    static const bool trace = [] { const char* e = getenv("HEXL_HOST_TRACE"); return e && atoi(e) == 1; }();
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace)
            fprintf(stderr, "[hexl host] %-22s zero-copy    +%8.1f us  @%12.1f us\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(),
                    fmod(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(), 1e8));
    };
    const size_t in_bytes = (batch * tt + 255) & ~size_t(255), out_bytes = (batch * rs + 255) & ~size_t(255);
    const size_t nflag = batch * 2 * L * 4;                                   // one word per quarter limb
    int rc = hx_reserve_pinned(c, &c->h_lone, &c->h_lone_bytes, in_bytes + out_bytes + 256 + nflag * sizeof(u32), true);
    if (rc) return rc;
    char* h_in = (char*)c->h_lone;
    char* h_out = h_in + in_bytes;
    u32* flag = (u32*)(h_out + out_bytes);
    u32* done = flag + 64;
    for (size_t b = 0; b < batch; ++b) memcpy(h_in + b * tt, h_t_targets[b], tt);
    __atomic_store_n(flag, 0u, __ATOMIC_RELAXED);                             // flag[0]: input-range flag, flag[1]: "the last kernel has started"
    __atomic_store_n(flag + 1, 0u, __ATOMIC_RELAXED);
    memset(done, 0, nflag * sizeof(u32));
    stamp("packed");
    const u32 epoch = ++c->lone_epoch ? c->lone_epoch : ++c->lone_epoch;      // never 0
    p->host_done = done; p->host_flag = flag; p->host_epoch = epoch; p->overwrite_result = true;
    rc = hexl_keyswitch(p, (u64*)h_out, (const u64*)h_in, batch);              // pinned slabs: device-visible at their host addresses
    p->host_done = nullptr; p->host_flag = nullptr; p->overwrite_result = false;
    if (rc) return rc;
    stamp("enqueued");
    bool distinct = true;                                                     // aliased results must be added in order
    for (size_t a = 0; a < batch && distinct; ++a)
        for (size_t b = a + 1; b < batch; ++b)
            if (h_results[a] == h_results[b]) { distinct = false; break; }
    auto add_limb = [&](u64* res, const u64* out, size_t limb) {                // limb = k * L + i
        const u64 q = p->moduli[limb % L];
        const u64* o = out + limb * n;
        u64* r = res + limb * n;
        hx_add_mod_u64(r, o, n, q);                                              // host_simd.cpp (AVX-512 / AVX2 / portable)
    };
    // a quarter limb has landed when its word shows this call's epoch (released by the kernel behind the data); a launch that died
    // never publishes: give up after the stream has gone idle without it
    // (spinning pool workers share the process-wide HostPool with other device runners: after a few thousand polls they yield)
    auto relax = [](unsigned spins) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0xFFF) == 0xFFF) std::this_thread::yield();
    };
    std::atomic<bool> lost{false};
    auto wait_limb = [&](size_t x) -> bool {                                    // x = b * 2L + limb
        for (int qd = 0; qd < 4; ++qd) {
            const u32* w = done + x * 4 + qd;
            for (unsigned spins = 0; __atomic_load_n(w, __ATOMIC_ACQUIRE) != epoch; ++spins) {
                if (lost.load(std::memory_order_relaxed)) return false;
                if ((spins & 0xFFFF) == 0xFFFF && hipStreamQuery(c->stream) != hipErrorNotReady &&
                    __atomic_load_n(w, __ATOMIC_ACQUIRE) != epoch) { lost.store(true); return false; }
                relax(spins);
            }
        }
        return true;
    };
    auto wait_word = [&](const u32* w) -> bool {
        for (unsigned spins = 0; __atomic_load_n(w, __ATOMIC_ACQUIRE) != epoch; ++spins) {
            if ((spins & 0xFFFF) == 0xFFFF && hipStreamQuery(c->stream) != hipErrorNotReady && __atomic_load_n(w, __ATOMIC_ACQUIRE) != epoch) {
                lost.store(true);
                return false;
            }
            relax(spins);
        }
        return true;
    };
    if (distinct) {
        // the calling thread alone until the LAST kernel starts (its arithmetic takes ~15 us before the first limb crosses PCIe) ...
        wait_word(flag + 1);
        stamp("last kernel started");
        // ... then one job per limb: the pool's workers wake up during that arithmetic and add limb x while limbs x + 1 ... are in flight
        parallel_for(batch * 2 * L, [&](size_t x) { if (wait_limb(x)) add_limb(h_results[x / (2 * L)], (const u64*)(h_out + (x / (2 * L)) * rs), x % (2 * L)); });
    } else {
        HX_CHECK(hipStreamSynchronize(c->stream));
        for (size_t b = 0; b < batch; ++b)
            for (size_t limb = 0; limb < 2 * L; ++limb) add_limb(h_results[b], (const u64*)(h_out + b * rs), limb);
    }
    if (lost.load()) {
        const hipError_t e = hipStreamSynchronize(c->stream);
        fprintf(stderr, "[hexl_mi355x] lone keyswitch: the launch finished without publishing its results (%s)\n", hipGetErrorString(e));
        return e != hipSuccess ? (int)e : (int)hipErrorUnknown;
    }
    // every quarter limb of the LAST kernel has been published, so the launch is over as far as these slabs and the plan's scratch
    // are concerned; the next launch on the stream is ordered behind it anyway
    stamp("done");
    return __atomic_load_n(flag, __ATOMIC_ACQUIRE) ? HEXL_W_RANGE : 0;
}

extern "C" int hexl_keyswitch_host(hexl_ks_plan* p, uint64_t* const* h_results, const uint64_t* const* h_t_targets,
                                   size_t batch) {
    if (!p || !h_results || !h_t_targets) return HEXL_E_BADARG;
    if (!batch) return 0;
    hexl_ctx* c = p->ctx;
    HX_CHECK(hipSetDevice(c->device));
    const size_t n = p->n, L = p->L;
    const size_t tt = L * n * 8, rs = 2 * tt;
    const int kind = payload_kind({{(const void* const*)h_results, batch}, {(const void* const*)h_t_targets, batch}});
    if (kind < 0) return HEXL_E_BADARG;
    if (kind == 1) {
        // device-resident ciphertexts: the kernels accumulate straight into `result`. A run ends where the arrays stop
        // being contiguous or a result array repeats (two instances of one launch must not update the same words;
        // launches on one stream are ordered, which keeps the reference's submission-order semantics)
        int rc = 0;
        // the status covers THIS call's objects, as on the staged path below: whatever an earlier hexl_keyswitch on the plan left in
        // the flag is for hexl_ks_range_check before this call (only the staged branch cleared it)
        if (p->use_f64) HX_CHECK(hipMemsetAsync(p->d_flag, 0, sizeof(u32), c->stream));
        for (size_t i = 0, j; !rc && i < batch; i = j) {
            for (j = i + 1; j < batch && h_results[j] == h_results[j - 1] + rs / 8 &&
                            h_t_targets[j] == h_t_targets[j - 1] + tt / 8; ++j) {}
            rc = hexl_keyswitch(p, h_results[i], h_t_targets[i], j - i);
        }
        // the status covers this call, as on the staged path below: the kernels' range flag is read (and cleared) here too
        if (!rc) rc = p->use_f64 ? hexl_ks_range_check(p) : (int)hipStreamSynchronize(c->stream);
        return rc;
    }
    // a lone keyswitch (or the few that still take the quarter-transform path): zero-copy, no staging pipeline. HEXL_HOST_ZERO_COPY=0
    // keeps the staged route (comparisons)
    static const bool zero_copy = [] { const char* e = getenv("HEXL_HOST_ZERO_COPY"); return !(e && atoi(e) == 0); }();
    // (one quarter-transform launch must cover the whole call -- its completion words and output gate are per launch: a batch that
    // HEXL_KS_CHUNK would cut into several chunks takes the staged route)
    if (zero_copy && p->use_f64 && p->have_keys && batch <= 8 && batch <= hx_ks_chunk(p) && hx_ks_lat_applies(p, batch) && hx_ks_can_overwrite(p, batch))
        return keyswitch_host_lone(p, h_results, h_t_targets, batch);
    // The FP64 kernels flag t_target words that are not below their modulus (the device-side result buffer starts at zero or
    // is written here, so `result` is the host's business). The status covers THIS call: the flag is cleared on the stream
    // before the first launch (whatever earlier hexl_keyswitch launches on the plan left in it is for hexl_ks_range_check,
    // before this call), every sub-batch's launch is followed by a 4-byte copy of the still accumulating flag into the pinned
    // mirror, and the pipeline's own last synchronisation covers the last copy -- no extra stream synchronisation per call.
    if (p->use_f64) { *p->h_flag = 0; HX_CHECK(hipMemsetAsync(p->d_flag, 0, sizeof(u32), c->stream)); }
    PipeShape sh{tt, rs, 0, sub_batch_for(tt + rs), false};
    // Like the reference, the device produces the keyswitch output only (the kernels accumulate into a zeroed
    // buffer) and the HOST adds it into the caller's result in submission order
    // (FPGAObject_KeySwitch::fill_out_data, fpga.cpp:441-475). This keeps the semantics when several objects of a
    // batch alias the same result array, as benchmark/bench_keyswitch.cpp:113-131 does; it also means `result`
    // never crosses PCIe upwards.
    auto add_limb = [&](u64* res, const u64* out, size_t limb) {  // limb = k * L + i
        const u64 q = p->moduli[limb % L];
        const u64* o = out + limb * n;
        u64* r = res + limb * n;
        hx_add_mod_u64(r, o, n, q);                                              // host_simd.cpp (AVX-512 / AVX2 / portable)
    };
    auto add_into = [&](u64* res, const u64* out) {
        for (size_t limb = 0; limb < 2 * L; ++limb) add_limb(res, out, limb);
    };
    const int rc_pipe = run_pipeline(c, batch, sh, [](char*) {},
        [&](size_t first, size_t cnt, char* h) {
            if (cnt >= 4) parallel_for(cnt, [&](size_t b) { memcpy(h + b * tt, h_t_targets[first + b], tt); });
            else for (size_t b = 0; b < cnt; ++b) memcpy(h + b * tt, h_t_targets[first + b], tt);
        },
        [&](size_t cnt, char* d, char* dout) {
            // small sub-batches run on kernels that can WRITE their output: no zeroed buffer to prepare and to read back
            const bool ow = hx_ks_can_overwrite(p, cnt);
            if (!ow) HX_CHECK(hipMemsetAsync(dout, 0, cnt * rs, c->stream));
            p->overwrite_result = ow;
            const int rc = hexl_keyswitch(p, (u64*)dout, (u64*)d, cnt);
            p->overwrite_result = false;
            if (!rc && p->use_f64) HX_CHECK(hipMemcpyAsync(p->h_flag, p->d_flag, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            return rc;
        },
        [&](size_t first, size_t cnt, const char* h) {
            bool distinct = true;                                   // aliased results must be added in order
            for (size_t a = 0; a < cnt && distinct; ++a)
                for (size_t b = a + 1; b < cnt; ++b)
                    if (h_results[first + a] == h_results[first + b]) { distinct = false; break; }
            // distinct results: one job per (item, limb) -- a lone keyswitch (worksize 1) still spreads over 2 L threads
            if (distinct)
                parallel_for(cnt * 2 * L, [&](size_t x) { add_limb(h_results[first + x / (2 * L)], (const u64*)(h + (x / (2 * L)) * rs), x % (2 * L)); });
            else for (size_t b = 0; b < cnt; ++b) add_into(h_results[first + b], (const u64*)(h + b * rs));
        });
    if (rc_pipe) return rc_pipe;
    return (p->use_f64 && *p->h_flag) ? HEXL_W_RANGE : 0;
}
