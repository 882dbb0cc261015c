// hexl_fpga_api.cpp -- libhexl-fpga.so: the reference's C++ API (include/hexl-fpga.h) on top of the
// C-ABI launcher library (include/hexl_mi355x.h). Plain C++17, no HIP headers.
//
// It replaces the reference's L3/L2 host layers: host/src/hexl-fpga.cpp (forwarding),
// host/src/{dyadic_multiply,keyswitch,ntt,intt}.cpp (argument checks), host/src/fpga_int.cpp (worksize,
// fence and completion logic :171-507) and the Buffer / Device::run / DevicePool machinery of
// host/src/fpga.cpp:100-180,780-866,1609-1685. Same shape as the reference:
//   * X() only appends an object to ONE submission-ordered FIFO (Buffer::push, fpga_int.cpp:420-462) and
//     returns; with worksize 1 it then waits for that object, like fpga_int.cpp:459-461.
//   * one RUNNER THREAD PER DEVICE (NUM_DEV, fpga.cpp:1646-1673) pops the head of the FIFO as soon as
//     something is there -- a maximal run of consecutive objects of one primitive with the same
//     parameters (the reference's batches never span a parameter change either: fences,
//     fpga_int.cpp:346-353,429-447) -- and pushes it through the device's staging pipeline while the caller keeps
//     enqueuing. Each device context is touched by its own runner only.
//   * XCompleted() waits on a condition variable until every object of that primitive is done (the reference
//     spins on ready_, fpga_int.cpp:484-507), returns true and resets the worksize to 1. Completion is tracked PER
//     OBJECT: every object carries a ticket of its primitive, a finished run retires its ticket range, and a waiter
//     (a worksize-1 call, XCompleted) returns when every ticket up to its own is retired -- with several devices
//     runs finish out of order, a bare counter would let a caller return while its own object is still in flight.
//   * a runner does not start on the first object of a batch window that is still being filled: it waits (bounded)
//     for its share -- worksize / devices -- or for the window to close (the worksize reached, XCompleted called, a
//     different parameter set queued behind the run).
//   * several devices take contiguous shares of a run. Objects whose OUTPUT array is still being produced on
//     another device wait for it (KeySwitch accumulates into `result`; benchmark/bench_keyswitch.cpp:113-131
//     submits the same result many times), and runs of different primitives never overlap, so the results are
//     those of the reference's in-order queue.
//   * keys / twiddles are cached per device and keyed by the parameter values and the k_switch_keys pointer
//     values (fpga.cpp:1158-1165); consecutive objects are matched by pointer identity first, like
//     fpga_int.cpp:429-447, so a steady stream of KeySwitch calls does not copy or compare parameter vectors.
#include "../../include/hexl-fpga.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <unordered_map>
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

// HEXL_HOST_TRACE=1: microseconds (steady clock, since the process's first stamp) at the hand-over points of a call, on stderr
bool host_trace() { static const bool on = env_ul("HEXL_HOST_TRACE", 0) == 1; return on; }
void trace_stamp(const char* what) {
    if (!host_trace()) return;
    // (absolute steady-clock microseconds modulo 10^8: the same time base as the staging pipeline's stamps in libhexl_mi355x.so)
    std::fprintf(stderr, "[hexl api ] %-28s @%12.1f us\n", what,
                 std::fmod(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(), 1e8));
}

enum Kind { DY = 0, KS = 1, NTT = 2, INTT = 3, NKIND = 4 };
const char* kind_name[NKIND] = {"DyadicMultiply", "KeySwitch", "NTT", "INTT"};

struct KsKey {   // identifies a device-side plan: parameter VALUES + key pointer identity + a fingerprint of the keys
    uint64_t n, L, K, rns, fp;
    std::vector<uint64_t> moduli, msf;
    std::vector<const uint64_t*> keys;
    const uint64_t* twiddles;
    bool operator<(const KsKey& o) const {
        return std::tie(n, L, K, rns, fp, moduli, msf, keys, twiddles) <
               std::tie(o.n, o.L, o.K, o.rns, o.fp, o.moduli, o.msf, o.keys, o.twiddles);
    }
};

// The reference's device key cache is keyed by the k_switch_keys pointer alone (Device::KeySwitch_check_keys,
// fpga.cpp:1158-1165): keys re-allocated at a recycled address with new contents silently hit the stale entry. A
// fingerprint of a few words of every key limb (first and last four of each key component) makes that case a miss.
uint64_t key_fingerprint(const uint64_t** keys, uint64_t L, uint64_t K, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; };
    for (uint64_t d = 0; d < L; ++d) {
        const uint64_t* k = keys[d];
        for (uint64_t c = 0; c < 2; ++c) {
            const uint64_t* comp = k + c * K * n;
            for (int j = 0; j < 4; ++j) { mix(comp[j]); mix(comp[K * n - 1 - j]); }
        }
    }
    return h;
}

// one queued call. `out` is the array the object writes (the aliasing rule looks at it); p0..p3 and s0..s3 are the
// call's remaining pointers / scalars in argument order; plan = index into Engine::ks_keys for KeySwitch.
struct Obj {
    Kind kind;
    uint64_t* out;
    const uint64_t *p0, *p1, *p2;
    uint64_t n, s0, s1, s2;
    int plan;
    uint64_t ticket = 0;         // 1-based sequence number among the objects of this primitive (set by submit)
};

struct Device {
    hexl_ctx* ctx = nullptr;
    // by Engine::ks_keys index; touched by this device's runner only. Bounded (HEXL_PLAN_CACHE, default 8 parameter/key
    // sets per device): a long-running application that rotates keys would otherwise pin tables, three key copies and
    // scratch for every key set it has ever used -- the least recently used plan is destroyed when a new one is needed.
    std::map<int, hexl_ks_plan*> plans;
    std::vector<int> plan_lru;                 // most recent last
    std::thread runner;
};

struct Engine {
    std::vector<Device> devs;
    int debug = 0;
    size_t max_run = 1024;          // FPGA_BUFSIZE: most objects one runner takes at a time
    std::mutex mu;                  // guards everything below
    std::condition_variable cv_work, cv_done;
    std::deque<Obj> fifo;
    size_t max_plans = 8;           // HEXL_PLAN_CACHE: device-side keyswitch plans kept per device
    uint64_t ws[NKIND] = {1, 1, 1, 1}, submitted[NKIND] = {0, 0, 0, 0};
    // per-object completion: retired[k] = every ticket <= it is done; finished ranges beyond it wait in `done_ranges`
    uint64_t retired[NKIND] = {0, 0, 0, 0};
    std::map<uint64_t, uint64_t> done_ranges[NKIND];             // first ticket -> last ticket of a finished run
    uint64_t window_open[NKIND] = {0, 0, 0, 0};                  // objects the current worksize window still expects
    int closers[NKIND] = {0, 0, 0, 0};                           // XCompleted() callers waiting: the window is closed
    int running_kind = -1, running_runs = 0;                     // runs in flight (all of one primitive)
    std::unordered_map<const void*, int> inflight_out;           // output arrays of the runs in flight
    bool stop = false;
    // keyswitch parameter sets seen so far + the pointer identity of the last call (fast path)
    std::vector<KsKey> ks_keys;
    std::map<KsKey, int> ks_index;
    struct { const uint64_t *moduli, *msf, *twiddles; const uint64_t** keys; uint64_t n, L, K, rns, fp; int plan; } last_ks = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, -1};
};

Engine* g = nullptr;

Engine& eng() {
    REQUIRE(g != nullptr, "acquire_FPGA_resources() must be called first");
    return *g;
}

bool same_params(const Obj& a, const Obj& b) {
    if (a.kind != b.kind) return false;
    switch (a.kind) {
        case DY:   return a.n == b.n && a.s0 == b.s0;                        // n, n_moduli
        case KS:   return a.plan == b.plan;
        case NTT:  return a.n == b.n && a.s0 == b.s0;                        // n, modulus (fpga_int.cpp:346-353)
        case INTT: return a.n == b.n && a.s0 == b.s0;
        default:   return false;
    }
}

hexl_ks_plan* plan_for(Engine& e, Device& dev, int idx) {
    auto touch = [&] {
        auto& l = dev.plan_lru;
        l.erase(std::remove(l.begin(), l.end(), idx), l.end());
        l.push_back(idx);
    };
    auto it = dev.plans.find(idx);
    if (it != dev.plans.end()) { touch(); return it->second; }
    KsKey k;
    size_t keep;
    { std::lock_guard<std::mutex> lk(e.mu); k = e.ks_keys[idx]; keep = e.max_plans; }
    while (dev.plans.size() >= keep && !dev.plan_lru.empty()) {   // evict the least recently used plan of this device
        const int old = dev.plan_lru.front();
        dev.plan_lru.erase(dev.plan_lru.begin());
        auto o = dev.plans.find(old);
        if (o != dev.plans.end()) { hexl_ks_plan_destroy(o->second); dev.plans.erase(o); }
    }
    hexl_ks_plan* p = nullptr;
    int rc = hexl_ks_plan_create(dev.ctx, k.n, k.L, k.K, k.rns, 2, k.moduli.data(), k.msf.data(), k.twiddles, &p);
    if (rc) die("hexl_ks_plan_create (unsupported keyswitch parameters?)", rc);
    rc = hexl_ks_set_keys(p, k.keys.data());
    if (rc) die("hexl_ks_set_keys", rc);
    dev.plans.emplace(idx, p);
    touch();
    return p;
}

void execute(Engine& e, Device& dev, const std::vector<Obj>& run) {
    const Obj& f = run.front();
    const size_t cnt = run.size();
    int rc = 0;
    switch (f.kind) {
        case DY: {
            std::vector<uint64_t*> out(cnt); std::vector<const uint64_t*> a(cnt), b(cnt), m(cnt);
            for (size_t k = 0; k < cnt; ++k) { out[k] = run[k].out; a[k] = run[k].p0; b[k] = run[k].p1; m[k] = run[k].p2; }
            rc = hexl_dyadic_multiply_host(dev.ctx, out.data(), a.data(), b.data(), cnt, f.n, m.data(), f.s0);
            break;
        }
        case KS: {
            std::vector<uint64_t*> r(cnt); std::vector<const uint64_t*> t(cnt);
            for (size_t k = 0; k < cnt; ++k) { r[k] = run[k].out; t[k] = run[k].p0; }
            rc = hexl_keyswitch_host(plan_for(e, dev, f.plan), r.data(), t.data(), cnt);
            break;
        }
        case NTT: {
            std::vector<uint64_t*> x(cnt);
            for (size_t k = 0; k < cnt; ++k) x[k] = run[k].out;
            // tables of the run's first object, like FPGAObject_NTT::fill_in_data (fpga.cpp:403-411)
            rc = hexl_ntt_fwd_host(dev.ctx, x.data(), cnt, f.p0, f.p1, f.s0, f.n);
            break;
        }
        case INTT: {
            std::vector<uint64_t*> x(cnt);
            for (size_t k = 0; k < cnt; ++k) x[k] = run[k].out;
            rc = hexl_ntt_inv_host(dev.ctx, x.data(), cnt, f.p0, f.p1, f.s0, f.s1, f.s2, f.n);
            break;
        }
        default: break;
    }
    if (rc == HEXL_W_RANGE && f.kind == KS) {
        // the run was computed, but one or more of its objects had a t_target word not below its modulus: outside
        // intel::hexl::KeySwitch's contract (the reference checks nothing and returns whatever its pipeline makes of it). Like an
        // FPGA_ASSERT (fpga_assert.h:24-38) this is fatal only under FPGA_DEBUG; otherwise count it, and say so the first time and
        // every 1000th. (HEXL_E_RANGE -- the HEXL_KS_VALIDATE=1 refusal, nothing was computed -- stays fatal below.)
        if (e.debug) die("KeySwitch: a t_target word is not below its modulus", rc);
        static std::atomic<unsigned long> seen{0};
        const unsigned long k = ++seen;
        if (k == 1 || k % 1000 == 0)
            std::fprintf(stderr, "[hexl-fpga/mi355x] warning: KeySwitch got a t_target word >= its modulus in one or more objects of a run of "
                                 "%zu (the results of those objects are unspecified); %lu such run(s) so far\n", cnt, k);
        return;
    }
    if (rc) die(kind_name[f.kind], rc);
}

// the runner of device `di` (Device::run, fpga.cpp:780-866)
void runner_loop(Engine* ep, size_t di) {
    Engine& e = *ep;
    Device& dev = e.devs[di];
    std::vector<Obj> run;
    std::unique_lock<std::mutex> lk(e.mu);
    const size_t nd = e.devs.size();
    for (;;) {
        run.clear();
        // the head run: same primitive, same parameters, no output another device is still producing
        auto head_run = [&]() -> size_t {
            if (e.fifo.empty()) return 0;
            const Obj& h = e.fifo.front();
            if (e.running_runs > 0 && e.running_kind != (int)h.kind) return 0;          // primitives never overlap
            size_t avail = 0;
            while (avail < e.fifo.size() && avail < e.max_run && same_params(h, e.fifo[avail]) &&
                   e.inflight_out.find(e.fifo[avail].out) == e.inflight_out.end())
                ++avail;
            return avail;
        };
        size_t avail = 0, take = 0;
        for (bool timed_out = false;;) {
            if (e.stop) return;
            avail = head_run();
            if (avail) {
                const Kind k = e.fifo.front().kind;
                // this runner's share of the batch window: worksize / devices (everything, for one device)
                const uint64_t ws = e.ws[k];
                const size_t share = std::min<size_t>(e.max_run, nd > 1 ? (size_t)((ws + nd - 1) / nd) : (size_t)ws);
                // closed: nothing more will join this run -- the window is full, XCompleted() is waiting, the run is cut short
                // by an object it cannot absorb, or the caller went quiet for longer than the bounded wait below
                const bool closed = e.window_open[k] == 0 || e.closers[k] > 0 || avail < e.fifo.size() || avail >= e.max_run || timed_out;
                if (avail >= share) { take = nd > 1 ? share : avail; break; }
                if (closed) {
                    // what is left of a window (less than a share): split it between the runners that are idle right now -- the
                    // busy ones have their shares -- but not into crumbs (a run has a fixed cost of a few hundred microseconds:
                    // staging pipeline, launches, synchronisation)
                    const size_t idle = nd > (size_t)e.running_runs ? nd - (size_t)e.running_runs : 1;
                    take = nd > 1 ? std::min(avail, std::max<size_t>(4, (avail + idle - 1) / idle)) : avail;
                    break;
                }
                // a window that is still being filled: objects arrive microseconds apart, wait for the share (bounded)
                // (system_clock: pthread_cond_timedwait, which ThreadSanitizer intercepts; wait_for's steady clock is not)
                timed_out = e.cv_work.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(500)) ==
                            std::cv_status::timeout;
                continue;
            }
            timed_out = false;
            e.cv_work.wait(lk);
        }
        for (size_t k = 0; k < take; ++k) { run.push_back(e.fifo.front()); e.fifo.pop_front(); }
        for (const Obj& o : run) ++e.inflight_out[o.out];
        e.running_kind = (int)run.front().kind;
        ++e.running_runs;
        if (e.debug)
            std::fprintf(stderr, "[hexl-fpga/mi355x] device %zu: %zu x %s (n = %lu), %zu still queued\n", di, run.size(),
                         kind_name[run.front().kind], (unsigned long)run.front().n, e.fifo.size());
        lk.unlock();
        if (nd > 1) e.cv_work.notify_all();                       // the rest of the run is for the other runners
        trace_stamp("runner: run taken");
        execute(e, dev, run);
        trace_stamp("runner: run executed");
        lk.lock();
        for (const Obj& o : run) {
            auto it = e.inflight_out.find(o.out);
            if (--it->second == 0) e.inflight_out.erase(it);
        }
        // retire the run's tickets (contiguous: a run is consecutive objects of one primitive in submission order)
        const Kind k = run.front().kind;
        e.done_ranges[k][run.front().ticket] = run.back().ticket;
        for (auto it = e.done_ranges[k].begin(); it != e.done_ranges[k].end() && it->first == e.retired[k] + 1;
             it = e.done_ranges[k].erase(it))
            e.retired[k] = it->second;
        --e.running_runs;
        e.cv_done.notify_all();
        e.cv_work.notify_all();
    }
}

// append one object; worksize 1 => wait for it (fpga_int.cpp:459-461) -- for ITS ticket, not for a count of finished objects
void submit(Engine& e, Obj o) {
    std::unique_lock<std::mutex> lk(e.mu);
    o.ticket = ++e.submitted[o.kind];
    const uint64_t mine = o.ticket;
    e.fifo.push_back(o);
    if (e.window_open[o.kind]) --e.window_open[o.kind];
    e.cv_work.notify_all();
    if (e.ws[o.kind] == 1) {
        trace_stamp("caller: submitted, waiting");
        e.cv_done.wait(lk, [&] { return e.retired[o.kind] >= mine; });
        trace_stamp("caller: woken");
    }
}

bool completed(Kind k) {
    Engine& e = eng();
    std::unique_lock<std::mutex> lk(e.mu);
    const uint64_t upto = e.submitted[k];
    ++e.closers[k];                                                // the window is closed: runners stop waiting for more
    e.cv_work.notify_all();
    e.cv_done.wait(lk, [&] { return e.retired[k] >= upto; });
    --e.closers[k];
    e.ws[k] = 1;                                                   // fpga_int.cpp:229,308,389,504
    e.window_open[k] = 0;
    return true;
}

void set_ws(Kind k, uint64_t ws) {
    Engine& e = eng();
    std::lock_guard<std::mutex> lk(e.mu);
    e.ws[k] = ws ? ws : 1;
    e.window_open[k] = e.ws[k] > 1 ? e.ws[k] : 0;                  // objects this window still expects
}

bool pow2_in(uint64_t n, uint64_t lo, uint64_t hi) { return n >= lo && n <= hi && (n & (n - 1)) == 0; }

}  // namespace

namespace intel {
namespace hexl {

void acquire_FPGA_resources() {
    if (g) return;
    Engine* e = new Engine();
    e->debug = (int)env_ul("FPGA_DEBUG", 0);
    e->max_run = env_ul("FPGA_BUFSIZE", 1024);
    if (e->max_run == 0) e->max_run = 1;
    e->max_plans = env_ul("HEXL_PLAN_CACHE", 8);
    if (e->max_plans == 0) e->max_plans = 1;
    // RUN_CHOICE (0 CPU / 1 emulator / 2 FPGA in the reference, fpga_int.cpp:40-60) has one meaning here:
    // the MI355X path. There is deliberately no CPU fallback.
    const unsigned long want = env_ul("NUM_DEV", 1);
    // HEXL_DEV_ALIAS=1 (tests on a one-GPU box): NUM_DEV contexts share the visible GPUs round robin
    const bool alias = env_ul("HEXL_DEV_ALIAS", 0) == 1;
    const int ngpu = hexl_device_count();
    for (unsigned long d = 0; d < (want ? want : 1); ++d) {
        Device dev;
        const int id = alias && ngpu > 0 ? (int)(d % (unsigned long)ngpu) : (int)d;
        int rc = hexl_ctx_create(id, &dev.ctx);
        if (rc) {
            if (d == 0) die("no MI355X device available (hexl_ctx_create)", rc);
            break;                              // fewer GPUs than NUM_DEV: use what exists
        }
        char buf[256];
        if (hexl_ctx_describe(dev.ctx, buf, sizeof(buf)) == 0) std::printf("%s\n", buf);
        e->devs.push_back(std::move(dev));
    }
    for (size_t d = 0; d < e->devs.size(); ++d) e->devs[d].runner = std::thread(runner_loop, e, d);
    g = e;
}

void release_FPGA_resources() {
    if (!g) return;
    for (int k = 0; k < NKIND; ++k) completed((Kind)k);           // drain what is still queued
    { std::lock_guard<std::mutex> lk(g->mu); g->stop = true; }
    g->cv_work.notify_all();
    for (auto& d : g->devs) d.runner.join();
    for (auto& d : g->devs) {
        for (auto& kv : d.plans) hexl_ks_plan_destroy(kv.second);
        hexl_ctx_destroy(d.ctx);
    }
    delete g;
    g = nullptr;
}

// ---------------------------------------------------------------- DyadicMultiply
void set_worksize_DyadicMultiply(uint64_t ws) { set_ws(DY, ws); }

void DyadicMultiply(uint64_t* results, const uint64_t* operand1, const uint64_t* operand2, uint64_t n,
                    const uint64_t* moduli, uint64_t n_moduli) {
    REQUIRE(results && operand1 && operand2 && moduli, "DyadicMultiply: null pointer");
    REQUIRE(n > 0, "DyadicMultiply: n must be a positive integer");                        // dyadic_multiply.cpp:19
    REQUIRE(n_moduli > 0, "DyadicMultiply: requires n_moduli > 0");
    submit(eng(), Obj{DY, results, operand1, operand2, moduli, n, n_moduli, 0, 0, -1});
}

bool DyadicMultiplyCompleted() { return completed(DY); }

// ---------------------------------------------------------------- KeySwitch
void set_worksize_KeySwitch(uint64_t ws) { set_ws(KS, ws); }

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
    Engine& e = eng();
    int plan;
    trace_stamp("caller: KeySwitch entered");
    for (uint64_t d = 0; d < decomp_modulus_size; ++d) REQUIRE(k_switch_keys[d], "KeySwitch: null key pointer");
    const uint64_t fp = key_fingerprint(k_switch_keys, decomp_modulus_size, key_modulus_size, n);
    {
        std::lock_guard<std::mutex> lk(e.mu);
        auto& l = e.last_ks;
        // same argument POINTERS and sizes as the previous call: same parameter set (fpga_int.cpp:429-447 compares
        // pointers too; the arrays are caller-owned and immutable until KeySwitchCompleted)
        if (l.plan >= 0 && l.moduli == moduli && l.msf == modswitch_factors && l.keys == k_switch_keys &&
            l.twiddles == twiddle_factors && l.n == n && l.L == decomp_modulus_size && l.K == key_modulus_size &&
            l.rns == rns_modulus_size && l.fp == fp && !e.ks_keys.empty() &&
            std::equal(e.ks_keys[l.plan].keys.begin(), e.ks_keys[l.plan].keys.end(), k_switch_keys)) {
            plan = l.plan;
        } else {
            KsKey k;
            k.n = n; k.L = decomp_modulus_size; k.K = key_modulus_size; k.rns = rns_modulus_size; k.fp = fp;
            k.moduli.assign(moduli, moduli + key_modulus_size);
            k.msf.assign(modswitch_factors, modswitch_factors + key_modulus_size);
            k.keys.assign(k_switch_keys, k_switch_keys + decomp_modulus_size);
            k.twiddles = twiddle_factors;
            auto it = e.ks_index.find(k);
            if (it == e.ks_index.end()) {
                it = e.ks_index.emplace(k, (int)e.ks_keys.size()).first;
                e.ks_keys.push_back(k);
            }
            plan = it->second;
            l = {moduli, modswitch_factors, twiddle_factors, k_switch_keys, n, decomp_modulus_size, key_modulus_size,
                 rns_modulus_size, fp, plan};
        }
    }
    submit(e, Obj{KS, result, t_target_iter_ptr, nullptr, nullptr, n, 0, 0, 0, plan});
}

bool KeySwitchCompleted() { return completed(KS); }

// ---------------------------------------------------------------- _NTT / _INTT
void _set_worksize_NTT(uint64_t ws) { set_ws(NTT, ws); }

void _NTT(uint64_t* operand, const uint64_t* root_of_unity_powers, const uint64_t* precon_root_of_unity_powers,
          uint64_t coeff_modulus, uint64_t n) {
    REQUIRE(operand && root_of_unity_powers && precon_root_of_unity_powers, "_NTT: null pointer");
    REQUIRE(pow2_in(n, 1024, 32768), "_NTT: requires n = 16384 (1024..32768 accepted here)");   // ntt.cpp:24
    submit(eng(), Obj{NTT, operand, root_of_unity_powers, precon_root_of_unity_powers, nullptr, n, coeff_modulus, 0, 0, -1});
}

bool _NTTCompleted() { return completed(NTT); }

void _set_worksize_INTT(uint64_t ws) { set_ws(INTT, ws); }

void _INTT(uint64_t* operand, const uint64_t* inv_root_of_unity_powers,
           const uint64_t* precon_inv_root_of_unity_powers, uint64_t coeff_modulus, uint64_t inv_n, uint64_t inv_n_w,
           uint64_t n) {
    REQUIRE(operand && inv_root_of_unity_powers && precon_inv_root_of_unity_powers, "_INTT: null pointer");
    REQUIRE(pow2_in(n, 1024, 32768), "_INTT: requires n = 16384 (1024..32768 accepted here)");  // intt.cpp:25
    submit(eng(), Obj{INTT, operand, inv_root_of_unity_powers, precon_inv_root_of_unity_powers, nullptr, n, coeff_modulus,
                      inv_n, inv_n_w, -1});
}

bool _INTTCompleted() { return completed(INTT); }

}  // namespace hexl
}  // namespace intel
