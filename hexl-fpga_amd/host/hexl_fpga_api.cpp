// hexl_fpga_api.cpp -- libhexl-fpga.so: the reference's C++ API (include/hexl-fpga.h) on top of the
// C-ABI launcher library (include/hexl_mi355x.h). Plain C++17, no HIP headers.
//
// It replaces the reference's L3/L2 host layers: host/src/hexl-fpga.cpp (forwarding),
// host/src/{dyadic_multiply,keyswitch,ntt,intt}.cpp (argument checks), host/src/fpga_int.cpp (worksize,
// fence and completion logic :171-507) and the Buffer / Device::run / DevicePool machinery of
// host/src/fpga.cpp:100-180,780-866,1609-1685. Semantics kept:
//   * one FIFO per primitive; a batch never spans a change of op parameters (fence, fpga_int.cpp:346-353,
//     429-447) -- consecutive compatible objects are launched together;
//   * worksize 1 => the call returns after completion; XCompleted() drains, returns true, resets ws to 1;
//   * NUM_DEV devices each take a contiguous share of a batch (the reference: one runner thread per board
//     popping the shared queue, fpga.cpp:1664-1672); keys/twiddles are cached per device and keyed by the
//     k_switch_keys pointer values (fpga.cpp:1158-1165).
#include "../../include/hexl-fpga.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/hexl_mi355x.h"

namespace {

[[noreturn]] void die(const char* what, int rc = 0) {
    std::fprintf(stderr, "[hexl-fpga/mi355x] fatal: %s (status %d)\n", what, rc);
    std::abort();
}
#define REQUIRE(cond, msg) do { if (!(cond)) die(msg); } while (0)

unsigned long env_ul(const char* name, unsigned long dflt) {
    const char* e = std::getenv(name);
    return e ? std::strtoul(e, nullptr, 10) : dflt;
}

struct KsKey {   // identifies a device-side plan: parameters + key pointer identity
    uint64_t n, L, K, rns;
    std::vector<uint64_t> moduli, msf;
    std::vector<const uint64_t*> keys;
    const uint64_t* twiddles;
    bool operator<(const KsKey& o) const {
        return std::tie(n, L, K, rns, moduli, msf, keys, twiddles) <
               std::tie(o.n, o.L, o.K, o.rns, o.moduli, o.msf, o.keys, o.twiddles);
    }
    bool operator==(const KsKey& o) const { return !(*this < o) && !(o < *this); }
};

struct Device {
    hexl_ctx* ctx = nullptr;
    std::map<KsKey, hexl_ks_plan*> plans;
};

struct DyObj { uint64_t* out; const uint64_t *a, *b, *moduli; uint64_t n, nm; };
struct NttObj { uint64_t* x; const uint64_t *roots, *precon; uint64_t q, n; };
struct InttObj { uint64_t* x; const uint64_t *roots, *precon; uint64_t q, inv_n, inv_n_w, n; };
struct KsObj { uint64_t* result; const uint64_t* t; KsKey key; };

struct Engine {
    std::vector<Device> devs;
    int debug = 0;
    size_t bufsize = 1024;          // FPGA_BUFSIZE: flush when this many objects are queued
    std::mutex mu_dy, mu_ks, mu_ntt, mu_intt;
    uint64_t ws_dy = 1, ws_ks = 1, ws_ntt = 1, ws_intt = 1;
    std::vector<DyObj> q_dy;
    std::vector<NttObj> q_ntt;
    std::vector<InttObj> q_intt;
    std::vector<KsObj> q_ks;
};

Engine* g = nullptr;

Engine& eng() {
    REQUIRE(g != nullptr, "acquire_FPGA_resources() must be called first");
    return *g;
}

// run fn(dev_index, begin, end) over contiguous shares of [0, count) on every device
template <class Fn>
void shard(Engine& e, size_t count, Fn fn) {
    const size_t nd = std::min(e.devs.size(), count ? count : size_t(1));
    if (nd <= 1) { fn(0, 0, count); return; }
    std::vector<std::thread> th;
    const size_t per = (count + nd - 1) / nd;
    for (size_t d = 0; d < nd; ++d) {
        const size_t b = d * per, en = std::min(count, b + per);
        if (b >= en) break;
        th.emplace_back([=, &fn] { fn(d, b, en); });
    }
    for (auto& t : th) t.join();
}

void flush_dyadic(Engine& e) {
    auto q = std::move(e.q_dy);
    e.q_dy.clear();
    size_t i = 0;
    while (i < q.size()) {                     // maximal runs of identical (n, n_moduli)
        size_t j = i + 1;
        while (j < q.size() && q[j].n == q[i].n && q[j].nm == q[i].nm) ++j;
        shard(e, j - i, [&](size_t d, size_t b, size_t en) {
            std::vector<uint64_t*> out; std::vector<const uint64_t*> a, bb, m;
            for (size_t k = i + b; k < i + en; ++k) { out.push_back(q[k].out); a.push_back(q[k].a); bb.push_back(q[k].b); m.push_back(q[k].moduli); }
            int rc = hexl_dyadic_multiply_host(e.devs[d].ctx, out.data(), a.data(), bb.data(), out.size(), q[i].n, m.data(), q[i].nm);
            if (rc) die("hexl_dyadic_multiply_host", rc);
        });
        i = j;
    }
}

void flush_ntt(Engine& e) {
    auto q = std::move(e.q_ntt);
    e.q_ntt.clear();
    size_t i = 0;
    while (i < q.size()) {                     // fence on modulus change (fpga_int.cpp:346-353)
        size_t j = i + 1;
        while (j < q.size() && q[j].q == q[i].q && q[j].n == q[i].n) ++j;
        shard(e, j - i, [&](size_t d, size_t b, size_t en) {
            std::vector<uint64_t*> x;
            for (size_t k = i + b; k < i + en; ++k) x.push_back(q[k].x);
            // tables of the batch's first object, like FPGAObject_NTT::fill_in_data (fpga.cpp:403-411)
            int rc = hexl_ntt_fwd_host(e.devs[d].ctx, x.data(), x.size(), q[i].roots, q[i].precon, q[i].q, q[i].n);
            if (rc) die("hexl_ntt_fwd_host", rc);
        });
        i = j;
    }
}

void flush_intt(Engine& e) {
    auto q = std::move(e.q_intt);
    e.q_intt.clear();
    size_t i = 0;
    while (i < q.size()) {
        size_t j = i + 1;
        while (j < q.size() && q[j].q == q[i].q && q[j].n == q[i].n) ++j;
        shard(e, j - i, [&](size_t d, size_t b, size_t en) {
            std::vector<uint64_t*> x;
            for (size_t k = i + b; k < i + en; ++k) x.push_back(q[k].x);
            int rc = hexl_ntt_inv_host(e.devs[d].ctx, x.data(), x.size(), q[i].roots, q[i].precon, q[i].q, q[i].inv_n,
                                       q[i].inv_n_w, q[i].n);
            if (rc) die("hexl_ntt_inv_host", rc);
        });
        i = j;
    }
}

hexl_ks_plan* plan_for(Device& dev, const KsKey& k) {
    auto it = dev.plans.find(k);
    if (it != dev.plans.end()) return it->second;
    hexl_ks_plan* p = nullptr;
    int rc = hexl_ks_plan_create(dev.ctx, k.n, k.L, k.K, k.rns, 2, k.moduli.data(), k.msf.data(), k.twiddles, &p);
    if (rc) die("hexl_ks_plan_create (unsupported keyswitch parameters?)", rc);
    rc = hexl_ks_set_keys(p, k.keys.data());
    if (rc) die("hexl_ks_set_keys", rc);
    dev.plans.emplace(k, p);
    return p;
}

void flush_ks(Engine& e) {
    auto q = std::move(e.q_ks);
    e.q_ks.clear();
    size_t i = 0;
    while (i < q.size()) {                     // fence on parameter / key change (fpga_int.cpp:429-447)
        size_t j = i + 1;
        while (j < q.size() && q[j].key == q[i].key) ++j;
        shard(e, j - i, [&](size_t d, size_t b, size_t en) {
            std::vector<uint64_t*> r; std::vector<const uint64_t*> t;
            for (size_t k = i + b; k < i + en; ++k) { r.push_back(q[k].result); t.push_back(q[k].t); }
            int rc = hexl_keyswitch_host(plan_for(e.devs[d], q[i].key), r.data(), t.data(), r.size());
            if (rc) die("hexl_keyswitch_host", rc);
        });
        i = j;
    }
}

bool pow2_in(uint64_t n, uint64_t lo, uint64_t hi) { return n >= lo && n <= hi && (n & (n - 1)) == 0; }

}  // namespace

namespace intel {
namespace hexl {

void acquire_FPGA_resources() {
    if (g) return;
    Engine* e = new Engine();
    e->debug = (int)env_ul("FPGA_DEBUG", 0);
    e->bufsize = env_ul("FPGA_BUFSIZE", 1024);
    if (e->bufsize == 0) e->bufsize = 1;
    // RUN_CHOICE (0 CPU / 1 emulator / 2 FPGA in the reference, fpga_int.cpp:40-60) has one meaning here:
    // the MI355X path. There is deliberately no CPU fallback.
    const unsigned long want = env_ul("NUM_DEV", 1);
    for (unsigned long d = 0; d < (want ? want : 1); ++d) {
        Device dev;
        int rc = hexl_ctx_create((int)d, &dev.ctx);
        if (rc) {
            if (d == 0) die("no MI355X device available (hexl_ctx_create)", rc);
            break;                              // fewer GPUs than NUM_DEV: use what exists
        }
        char buf[256];
        if (hexl_ctx_describe(dev.ctx, buf, sizeof(buf)) == 0) std::printf("%s\n", buf);
        e->devs.push_back(dev);
    }
    g = e;
}

void release_FPGA_resources() {
    if (!g) return;
    for (auto& d : g->devs) {
        for (auto& kv : d.plans) hexl_ks_plan_destroy(kv.second);
        hexl_ctx_destroy(d.ctx);
    }
    delete g;
    g = nullptr;
}

// ---------------------------------------------------------------- DyadicMultiply
void set_worksize_DyadicMultiply(uint64_t ws) {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_dy);
    e.ws_dy = ws ? ws : 1;
}

void DyadicMultiply(uint64_t* results, const uint64_t* operand1, const uint64_t* operand2, uint64_t n,
                    const uint64_t* moduli, uint64_t n_moduli) {
    REQUIRE(results && operand1 && operand2 && moduli, "DyadicMultiply: null pointer");
    REQUIRE(n > 0, "DyadicMultiply: n must be a positive integer");                        // dyadic_multiply.cpp:19
    REQUIRE(n_moduli > 0, "DyadicMultiply: requires n_moduli > 0");
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_dy);
    e.q_dy.push_back({results, operand1, operand2, moduli, n, n_moduli});
    if (e.ws_dy == 1 || e.q_dy.size() >= e.bufsize) flush_dyadic(e);
}

bool DyadicMultiplyCompleted() {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_dy);
    flush_dyadic(e);
    e.ws_dy = 1;
    return true;
}

// ---------------------------------------------------------------- KeySwitch
void set_worksize_KeySwitch(uint64_t ws) {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_ks);
    e.ws_ks = ws ? ws : 1;
}

void KeySwitch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n, uint64_t decomp_modulus_size,
               uint64_t key_modulus_size, uint64_t rns_modulus_size, uint64_t key_component_count,
               const uint64_t* moduli, const uint64_t** k_switch_keys, const uint64_t* modswitch_factors,
               const uint64_t* twiddle_factors) {
    REQUIRE(result && t_target_iter_ptr && moduli && k_switch_keys && modswitch_factors, "KeySwitch: null pointer");
    REQUIRE(pow2_in(n, 1024, 32768), "KeySwitch: requires n = 32768/16384/8192/4096/2048/1024");   // keyswitch.cpp:23-26
    REQUIRE(decomp_modulus_size > 0 && rns_modulus_size > 0, "KeySwitch: requires decomp/rns modulus size > 0");
    REQUIRE(key_component_count == 2, "KeySwitch: requires key_component_count = 2");
    REQUIRE(decomp_modulus_size < key_modulus_size && key_modulus_size <= 16,
            "KeySwitch: requires decomp_modulus_size < key_modulus_size <= 16");            // reference: <= 7
    KsKey k;
    k.n = n; k.L = decomp_modulus_size; k.K = key_modulus_size; k.rns = rns_modulus_size;
    k.moduli.assign(moduli, moduli + key_modulus_size);
    k.msf.assign(modswitch_factors, modswitch_factors + key_modulus_size);
    k.keys.assign(k_switch_keys, k_switch_keys + decomp_modulus_size);
    k.twiddles = twiddle_factors;
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_ks);
    e.q_ks.push_back({result, t_target_iter_ptr, std::move(k)});
    if (e.ws_ks == 1 || e.q_ks.size() >= e.bufsize) flush_ks(e);
}

bool KeySwitchCompleted() {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_ks);
    flush_ks(e);
    e.ws_ks = 1;
    return true;
}

// ---------------------------------------------------------------- _NTT / _INTT
void _set_worksize_NTT(uint64_t ws) {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_ntt);
    e.ws_ntt = ws ? ws : 1;
}

void _NTT(uint64_t* operand, const uint64_t* root_of_unity_powers, const uint64_t* precon_root_of_unity_powers,
          uint64_t coeff_modulus, uint64_t n) {
    REQUIRE(operand && root_of_unity_powers && precon_root_of_unity_powers, "_NTT: null pointer");
    REQUIRE(pow2_in(n, 1024, 32768), "_NTT: requires n = 16384 (1024..32768 accepted here)");   // ntt.cpp:24
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_ntt);
    e.q_ntt.push_back({operand, root_of_unity_powers, precon_root_of_unity_powers, coeff_modulus, n});
    if (e.ws_ntt == 1 || e.q_ntt.size() >= e.bufsize) flush_ntt(e);
}

bool _NTTCompleted() {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_ntt);
    flush_ntt(e);
    e.ws_ntt = 1;
    return true;
}

void _set_worksize_INTT(uint64_t ws) {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_intt);
    e.ws_intt = ws ? ws : 1;
}

void _INTT(uint64_t* operand, const uint64_t* inv_root_of_unity_powers,
           const uint64_t* precon_inv_root_of_unity_powers, uint64_t coeff_modulus, uint64_t inv_n, uint64_t inv_n_w,
           uint64_t n) {
    REQUIRE(operand && inv_root_of_unity_powers && precon_inv_root_of_unity_powers, "_INTT: null pointer");
    REQUIRE(pow2_in(n, 1024, 32768), "_INTT: requires n = 16384 (1024..32768 accepted here)");  // intt.cpp:25
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_intt);
    e.q_intt.push_back({operand, inv_root_of_unity_powers, precon_inv_root_of_unity_powers, coeff_modulus, inv_n,
                        inv_n_w, n});
    if (e.ws_intt == 1 || e.q_intt.size() >= e.bufsize) flush_intt(e);
}

bool _INTTCompleted() {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu_intt);
    flush_intt(e);
    e.ws_intt = 1;
    return true;
}

}  // namespace hexl
}  // namespace intel
