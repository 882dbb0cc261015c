// fake_mi355x.cpp -- TEST INFRASTRUCTURE: a CPU stand-in for the C-ABI of include/hexl_mi355x.h, built as a separate
// libhexl_mi355x.so under tests/cpp/_host/ so that the REAL host layer (hexl-fpga_amd/host/hexl_fpga_api.cpp: FIFO,
// runner threads, fences, NUM_DEV sharding, completion) can be exercised without a GPU -- under ThreadSanitizer, with
// any number of "devices". Compute is the oracle (oracle/liborc.so). The product never sees this file: the shipped
// libhexl-fpga.so links hexl-fpga_amd/lib/libhexl_mi355x.so (tests/test_abi.py::test_product_never_touches_the_oracle).
//
// It also checks the host layer's side of the contract: a context is driven by one thread at a time (the ADVICE
// round-1 finding: several primitives flushing through one context concurrently), and a plan is used with the
// context it was created on.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/hexl_mi355x.h"
#include "../../oracle/hexl_oracle.h"

struct hexl_ctx {
    int device;
    std::atomic<int> busy{0};
};
struct hexl_ks_plan {
    hexl_ctx* ctx;
    uint64_t n, L, K, rns;
    std::vector<uint64_t> moduli, msf, twiddles;
    std::vector<std::vector<uint64_t>> keys;
};

namespace {
struct Busy {   // aborts if two threads are inside one context at once
    hexl_ctx* c;
    explicit Busy(hexl_ctx* ctx) : c(ctx) {
        if (c->busy.fetch_add(1) != 0) { std::fprintf(stderr, "fake_mi355x: context %d entered concurrently\n", c->device); std::abort(); }
    }
    ~Busy() { c->busy.fetch_sub(1); }
};
// FAKE_DELAY_US: make runs long enough for the caller / other runners to interleave; FAKE_DELAY_DEV0_X: device 0 that many
// times slower, so that runs complete OUT OF ORDER across devices (per-object completion, tests/cpp/test_cxx_api.cpp threads)
void work_delay(const hexl_ctx* c) {
    static const long us = [] { const char* e = std::getenv("FAKE_DELAY_US"); return e ? atol(e) : 0L; }();
    static const long x0 = [] { const char* e = std::getenv("FAKE_DELAY_DEV0_X"); return e ? atol(e) : 1L; }();
    if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us * (c->device == 0 ? x0 : 1)));
}
}  // namespace

extern "C" {

int hexl_device_count(void) {
    const char* e = std::getenv("FAKE_DEVICES");
    return e ? atoi(e) : 4;
}
int hexl_ctx_create(int device, hexl_ctx** out) {
    if (!out || device < 0 || device >= hexl_device_count()) return HEXL_E_NODEVICE;
    *out = new hexl_ctx();
    (*out)->device = device;
    return 0;
}
int hexl_ctx_destroy(hexl_ctx* c) { delete c; return 0; }
int hexl_ctx_set_stream(hexl_ctx*, void*) { return 0; }
int hexl_ctx_use_own_stream(hexl_ctx*) { return 0; }
int hexl_ctx_sync(hexl_ctx*) { return 0; }
int hexl_ctx_describe(hexl_ctx* c, char* buf, size_t len) {
    std::snprintf(buf, len, "fake_mi355x (CPU oracle, test only): device %d", c->device);
    return 0;
}

int hexl_ntt_fwd_host(hexl_ctx* c, uint64_t* const* x, size_t batch, const uint64_t* roots, const uint64_t* precon, uint64_t q,
                      uint64_t n) {
    Busy b(c);
    work_delay(c);                                                 // delay BEFORE the compute: a caller released early sees stale data
    for (size_t k = 0; k < batch; ++k) orc_ntt_fwd(x[k], n, q, roots, precon);
    return 0;
}
int hexl_ntt_inv_host(hexl_ctx* c, uint64_t* const* x, size_t batch, const uint64_t* ir, const uint64_t* ip, uint64_t q,
                      uint64_t inv_n, uint64_t inv_n_w, uint64_t n) {
    Busy b(c);
    work_delay(c);
    for (size_t k = 0; k < batch; ++k) orc_ntt_inv(x[k], n, q, ir, ip, inv_n, inv_n_w);
    return 0;
}
int hexl_dyadic_multiply_host(hexl_ctx* c, uint64_t* const* out, const uint64_t* const* a, const uint64_t* const* bb, size_t batch,
                              uint64_t n, const uint64_t* const* moduli, uint64_t nm) {
    Busy b(c);
    work_delay(c);
    for (size_t k = 0; k < batch; ++k) orc_dyadic_multiply(out[k], a[k], bb[k], n, moduli[k], nm, 1);
    return 0;
}

int hexl_ks_plan_create(hexl_ctx* c, uint64_t n, uint64_t L, uint64_t K, uint64_t rns, uint64_t kcc, const uint64_t* moduli,
                        const uint64_t* msf, const uint64_t* twiddles, hexl_ks_plan** out) {
    if (!c || !out || kcc != 2 || !L || L >= K) return HEXL_E_BADARG;
    Busy b(c);
    hexl_ks_plan* p = new hexl_ks_plan();
    p->ctx = c; p->n = n; p->L = L; p->K = K; p->rns = rns;
    p->moduli.assign(moduli, moduli + K);
    p->msf.assign(msf, msf + K);
    if (twiddles) p->twiddles.assign(twiddles, twiddles + K * 4 * n);
    *out = p;
    return 0;
}
int hexl_ks_plan_destroy(hexl_ks_plan* p) { delete p; return 0; }
int hexl_ks_set_keys(hexl_ks_plan* p, const uint64_t* const* keys) {
    Busy b(p->ctx);
    p->keys.clear();
    for (uint64_t d = 0; d < p->L; ++d) p->keys.emplace_back(keys[d], keys[d] + 2 * p->K * p->n);
    return 0;
}
int hexl_ks_range_check(hexl_ks_plan*) { return 0; }
int hexl_keyswitch_host(hexl_ks_plan* p, uint64_t* const* results, const uint64_t* const* ts, size_t batch) {
    if (p->keys.empty()) return HEXL_E_NOKEYS;
    Busy b(p->ctx);
    work_delay(p->ctx);
    std::vector<const uint64_t*> kp;
    for (auto& k : p->keys) kp.push_back(k.data());
    for (size_t k = 0; k < batch; ++k) {        // plain (non-atomic) read-modify-write of the caller's result, like the product
        int rc = orc_keyswitch(results[k], ts[k], p->n, p->L, p->K, p->rns, 2, p->moduli.data(), kp.data(), p->msf.data(),
                               p->twiddles.empty() ? nullptr : p->twiddles.data());
        if (rc) return HEXL_E_BADARG;
    }
    return 0;
}

}  // extern "C"
