// test_cxx_api.cpp -- exercises libhexl-fpga.so through the reference's public C++ API
// (include/hexl-fpga.h == host/inc/hexl-fpga.h signatures) the way the reference's gtests do
// (tests/test_fwd_ntt.cpp:27-58, test_inv_ntt.cpp, test_dyadic_multiply.cpp:87-147, test_keyswitch.cpp:122-146,
// test_dyadic_multiply_keyswitch.cpp:295-313) and checks every result bit-for-bit against the CPU oracle
// (oracle/liborc.so). Built by tests/cpp/Makefile, run on the GPU box by tests/test_gpu_cxx_api.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "../../include/hexl-fpga.h"
#include "../../oracle/hexl_oracle.h"

#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
using namespace intel::hexl;
typedef std::vector<uint64_t> vec;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

struct Tables {
    uint64_t n, q, w, inv_n, inv_n_w;
    vec roots, precon, iroots, iprecon;
    Tables(uint64_t n_, uint64_t q_) : n(n_), q(q_), roots(n_), precon(n_), iroots(n_), iprecon(n_) {
        w = orc_minimal_primitive_root(2 * n, q);
        orc_tables_hexl(n, q, w, roots.data(), precon.data(), iroots.data(), iprecon.data());
        inv_n = orc_invmod(n, q);
        inv_n_w = orc_mulmod(inv_n, iroots[n - 1], q);
    }
};

static void test_ntt_intt(uint64_t n, unsigned bits, size_t batch) {
    uint64_t q;
    orc_generate_primes(&q, 1, bits, n);
    Tables t(n, q);
    std::vector<vec> x(batch, vec(n)), ref(batch);
    for (size_t b = 0; b < batch; ++b) {
        orc_fill_splitmix(x[b].data(), n, 77 + b, b == 1 ? 0 : q);        // object 1: out-of-range 64-bit garbage
        ref[b] = x[b];
        orc_ntt_fwd(ref[b].data(), n, q, t.roots.data(), t.precon.data());
    }
    _set_worksize_NTT(batch);
    for (size_t b = 0; b < batch; ++b) _NTT(x[b].data(), t.roots.data(), t.precon.data(), q, n);
    CHECK(_NTTCompleted(), "_NTTCompleted");
    for (size_t b = 0; b < batch; ++b) CHECK(x[b] == ref[b], "fwd NTT n=%lu bits=%u obj %zu", n, bits, b);
    for (size_t b = 0; b < batch; ++b) orc_ntt_inv(ref[b].data(), n, q, t.iroots.data(), t.iprecon.data(), t.inv_n, t.inv_n_w);
    _set_worksize_INTT(batch);
    for (size_t b = 0; b < batch; ++b) _INTT(x[b].data(), t.iroots.data(), t.iprecon.data(), q, t.inv_n, t.inv_n_w, n);
    CHECK(_INTTCompleted(), "_INTTCompleted");
    for (size_t b = 0; b < batch; ++b) CHECK(x[b] == ref[b], "inv NTT n=%lu bits=%u obj %zu", n, bits, b);
    // worksize 1: synchronous
    vec y(n), r;
    orc_fill_splitmix(y.data(), n, 5, q);
    r = y;
    orc_ntt_fwd(r.data(), n, q, t.roots.data(), t.precon.data());
    _NTT(y.data(), t.roots.data(), t.precon.data(), q, n);
    CHECK(y == r, "fwd NTT ws=1");
}

static void test_dyadic(uint64_t n, uint64_t nm, size_t batch) {
    // the reference's stimulus: toy moduli (b+m+1)*10 and ramps (tests/test_dyadic_multiply.cpp:35-52)
    std::vector<vec> a(batch, vec(2 * nm * n)), b(batch, vec(2 * nm * n)), out(batch, vec(3 * nm * n)), mod(batch, vec(nm));
    for (size_t k = 0; k < batch; ++k) {
        for (uint64_t m = 0; m < nm; ++m) mod[k][m] = (k + m + 1) * 10;
        for (uint64_t m = 0; m < nm; ++m)
            for (uint64_t i = 0; i < n; ++i) {
                a[k][m * n + i] = k + i + 1 + m * n;        b[k][m * n + i] = k + i + 2 + m * n;
                a[k][(nm + m) * n + i] = k + i + 11 + m * n; b[k][(nm + m) * n + i] = k + i + 22 + m * n;
            }
    }
    set_worksize_DyadicMultiply(batch);
    for (size_t k = 0; k < batch; ++k) DyadicMultiply(out[k].data(), a[k].data(), b[k].data(), n, mod[k].data(), nm);
    CHECK(DyadicMultiplyCompleted(), "DyadicMultiplyCompleted");
    for (size_t k = 0; k < batch; ++k) {
        vec ref(3 * nm * n);
        orc_dyadic_multiply(ref.data(), a[k].data(), b[k].data(), n, mod[k].data(), nm, 1);
        CHECK(out[k] == ref, "dyadic n=%lu nm=%lu obj %zu", n, nm, k);
    }
}

struct KsSetup {
    uint64_t n, L, K;
    vec moduli, msf;
    std::vector<vec> keys;
    std::vector<const uint64_t*> key_ptrs;
    KsSetup(uint64_t n_, uint64_t L_, uint64_t K_, uint64_t seed) : n(n_), L(L_), K(K_), moduli(K_), msf(K_) {
        orc_generate_primes(moduli.data(), K, 51, n);
        for (uint64_t i = 0; i < K; ++i) msf[i] = i + 1 < K ? orc_invmod(moduli[K - 1] % moduli[i], moduli[i]) : 1;
        for (uint64_t d = 0; d < L; ++d) {
            keys.emplace_back(2 * K * n);
            for (uint64_t k = 0; k < 2; ++k)
                for (uint64_t i = 0; i < K; ++i) orc_fill_splitmix(&keys[d][(k * K + i) * n], n, seed * 1000 + d * 64 + k * 16 + i, moduli[i]);
        }
        for (auto& k : keys) key_ptrs.push_back(k.data());
    }
    void make(vec& t, vec& r, uint64_t s) const {
        t.resize(L * n); r.resize(2 * L * n);
        for (uint64_t d = 0; d < L; ++d) orc_fill_splitmix(&t[d * n], n, s * 100 + d, moduli[d]);
        for (uint64_t k = 0; k < 2; ++k)
            for (uint64_t i = 0; i < L; ++i) orc_fill_splitmix(&r[(k * L + i) * n], n, s * 100 + 50 + k * 8 + i, moduli[i]);
    }
    void expect(vec& r, const vec& t) const {
        std::vector<const uint64_t*> kp(key_ptrs);
        orc_keyswitch(r.data(), t.data(), n, L, K, L + 1, 2, moduli.data(), kp.data(), msf.data(), nullptr);
    }
};

static void test_keyswitch(uint64_t n, uint64_t L, uint64_t K, size_t batch) {
    KsSetup ks(n, L, K, n + L);
    std::vector<vec> t(batch), r(batch), ref(batch);
    for (size_t b = 0; b < batch; ++b) { ks.make(t[b], r[b], b + 1); ref[b] = r[b]; ks.expect(ref[b], t[b]); }
    set_worksize_KeySwitch(batch);
    for (size_t b = 0; b < batch; ++b)
        KeySwitch(r[b].data(), t[b].data(), n, L, K, L + 1, 2, ks.moduli.data(), ks.key_ptrs.data(), ks.msf.data());
    CHECK(KeySwitchCompleted(), "KeySwitchCompleted");
    for (size_t b = 0; b < batch; ++b) CHECK(r[b] == ref[b], "keyswitch n=%lu L=%lu K=%lu obj %zu", n, L, K, b);
}

// several objects of one batch alias the same result array (benchmark/bench_keyswitch.cpp:113-131 submits the
// same vectors n_iter times in one worksize window): outputs must accumulate in submission order
static void test_keyswitch_aliased_results() {
    KsSetup ks(8192, 3, 4, 77);
    vec t, r, ref;
    ks.make(t, r, 5);
    ref = r;
    for (int it = 0; it < 3; ++it) ks.expect(ref, t);
    set_worksize_KeySwitch(3);
    for (int it = 0; it < 3; ++it)
        KeySwitch(r.data(), t.data(), ks.n, ks.L, ks.K, ks.L + 1, 2, ks.moduli.data(), ks.key_ptrs.data(), ks.msf.data());
    KeySwitchCompleted();
    CHECK(r == ref, "aliased result arrays inside one batch");
}

// benchmark/bench_keyswitch.cpp:113-131: ITER x (two test vectors) in ONE worksize window, every iteration accumulating
// into the same two result arrays. With HEXL_HOST_SUB_MB=1 every object is its own staging sub-batch, so the
// accumulating unpacks of consecutive sub-batches (and, with NUM_DEV > 1, of different devices) must be ordered.
// KeySwitch adds out(t) to result, so after `iters` rounds result = r0 + iters * out(t) (mod q_i).
static void test_keyswitch_aliased_across_subbatches(int repeats) {
    const int iters = 40;
    KsSetup ks(8192, 3, 4, 91);
    vec t[2], r0[2], out[2];
    for (int v = 0; v < 2; ++v) {
        ks.make(t[v], r0[v], 20 + v);
        out[v].assign(r0[v].size(), 0);
        ks.expect(out[v], t[v]);                                  // out(t): the oracle on a zero result
    }
    for (int rep = 0; rep < repeats; ++rep) {
        vec r[2] = {r0[0], r0[1]};
        set_worksize_KeySwitch(2 * iters);
        for (int it = 0; it < iters; ++it)
            for (int v = 0; v < 2; ++v)
                KeySwitch(r[v].data(), t[v].data(), ks.n, ks.L, ks.K, ks.L + 1, 2, ks.moduli.data(), ks.key_ptrs.data(), ks.msf.data());
        KeySwitchCompleted();
        for (int v = 0; v < 2; ++v) {
            bool ok = true;
            for (uint64_t k = 0; k < 2 && ok; ++k)
                for (uint64_t i = 0; i < ks.L && ok; ++i) {
                    const uint64_t q = ks.moduli[i];
                    for (uint64_t j = 0; j < ks.n; ++j) {
                        const size_t at = (k * ks.L + i) * ks.n + j;
                        const uint64_t want = (uint64_t)(((unsigned __int128)out[v][at] * iters + r0[v][at]) % q);
                        if (r[v][at] != want) { ok = false; break; }
                    }
                }
            CHECK(ok, "aliased results across sub-batches: repeat %d vector %d", rep, v);
            if (!ok) return;
        }
    }
}

// mixed op types and parameter changes inside one worksize window (fences), like
// tests/test_dyadic_multiply_keyswitch.cpp:295-313
static void test_mixed_and_fences() {
    KsSetup k1(4096, 2, 3, 1), k2(2048, 3, 4, 2);
    vec t1, r1, t2, r2, e1, e2;
    k1.make(t1, r1, 9); k2.make(t2, r2, 9);
    e1 = r1; e2 = r2; k1.expect(e1, t1); k2.expect(e2, t2);
    set_worksize_KeySwitch(2);
    KeySwitch(r1.data(), t1.data(), k1.n, k1.L, k1.K, k1.L + 1, 2, k1.moduli.data(), k1.key_ptrs.data(), k1.msf.data());
    test_dyadic(1024, 2, 3);                                   // a different primitive in between
    KeySwitch(r2.data(), t2.data(), k2.n, k2.L, k2.K, k2.L + 1, 2, k2.moduli.data(), k2.key_ptrs.data(), k2.msf.data());
    KeySwitchCompleted();
    CHECK(r1 == e1 && r2 == e2, "keyswitch with parameter change inside one worksize window");
    // NTT objects with two different moduli in one window
    uint64_t q[2];
    orc_generate_primes(q, 2, 40, 16384);
    Tables ta(16384, q[0]), tb(16384, q[1]);
    vec xa(16384), xb(16384), ea, eb;
    orc_fill_splitmix(xa.data(), 16384, 1, q[0]); orc_fill_splitmix(xb.data(), 16384, 2, q[1]);
    ea = xa; eb = xb;
    orc_ntt_fwd(ea.data(), 16384, q[0], ta.roots.data(), ta.precon.data());
    orc_ntt_fwd(eb.data(), 16384, q[1], tb.roots.data(), tb.precon.data());
    _set_worksize_NTT(2);
    _NTT(xa.data(), ta.roots.data(), ta.precon.data(), q[0], 16384);
    _NTT(xb.data(), tb.roots.data(), tb.precon.data(), q[1], 16384);
    _NTTCompleted();
    CHECK(xa == ea && xb == eb, "NTT fence on modulus change");
}

// Several caller threads, worksize 1 each (what concurrent SEAL evaluators do): X() must not return before ITS object is
// done. With several devices runs finish out of order; a completion COUNTER would release a caller whose own object is
// still running on a slower device (per-object tickets: hexl_fpga_api.cpp submit / runner_loop).
#include <thread>
static void test_concurrent_worksize1(int threads, int rounds) {
    uint64_t q;
    orc_generate_primes(&q, 1, 50, 16384);
    Tables t(16384, q);
    std::vector<int> bad(threads, 0);
    std::vector<std::thread> th;
    for (int id = 0; id < threads; ++id)
        th.emplace_back([&, id] {
            vec x(16384), ref;
            for (int r = 0; r < rounds; ++r) {
                orc_fill_splitmix(x.data(), 16384, 1000 * id + r, q);
                ref = x;
                orc_ntt_fwd(ref.data(), 16384, q, t.roots.data(), t.precon.data());
                _NTT(x.data(), t.roots.data(), t.precon.data(), q, 16384);     // worksize 1: done when it returns
                if (x != ref) ++bad[id];
            }
        });
    for (auto& x : th) x.join();
    for (int id = 0; id < threads; ++id) CHECK(bad[id] == 0, "thread %d saw %d unfinished worksize-1 results", id, bad[id]);
}

// Randomised windows through the host-pointer API (round 6, after the soaks of tools/ found a one-in-10^4 race in the inverse kernels that
// the small fixed windows above could not see): random primitive, ring dimension, modulus size and worksize -- up to hundreds of objects,
// i.e. persistent kernels, several staging sub-batches and many small workgroups per CU --, objects drawn from three distinct inputs,
// EVERY object of every window compared with the oracle.
static uint64_t rnd_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 7; rnd_state ^= rnd_state << 17; return rnd_state; }
static void stress(double seconds, uint64_t seed) {
    rnd_state ^= seed * 0x2545F4914F6CDD1Dull;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    size_t windows = 0, objects = 0;
    const uint64_t sizes[] = {1024, 2048, 2048, 4096, 8192, 16384};
    const size_t works[] = {1, 2, 3, 7, 16, 33, 64, 130, 257, 600};
    while (elapsed() < seconds) {
        const uint64_t n = sizes[rnd() % 6];
        size_t ws = works[rnd() % 10];
        const int prim = int(rnd() % 3);                            // 0: keyswitch, 1: _NTT then _INTT, 2: dyadic
        if (prim == 0) {
            const uint64_t L = 1 + rnd() % 3, K = L + 1;
            ws = std::min<size_t>(ws, n >= 16384 ? 130 : 600);
            KsSetup ks(n, L, K, 11 + rnd() % 1000);
            vec t[3], r[3], ref[3];
            for (int v = 0; v < 3; ++v) { ks.make(t[v], r[v], 3 + v + rnd() % 50); ref[v] = r[v]; ks.expect(ref[v], t[v]); }
            std::vector<vec> rr(ws);
            for (size_t b = 0; b < ws; ++b) rr[b] = r[b % 3];
            set_worksize_KeySwitch(ws);
            for (size_t b = 0; b < ws; ++b)
                KeySwitch(rr[b].data(), t[b % 3].data(), n, L, K, L + 1, 2, ks.moduli.data(), ks.key_ptrs.data(), ks.msf.data());
            CHECK(KeySwitchCompleted(), "KeySwitchCompleted");
            for (size_t b = 0; b < ws; ++b) CHECK(rr[b] == ref[b % 3], "stress keyswitch n=%lu L=%lu worksize %zu obj %zu", n, L, ws, b);
        } else if (prim == 1) {
            const unsigned bits = 25 + unsigned(rnd() % 36);
            uint64_t q;
            orc_generate_primes(&q, 1, bits, n);
            Tables tb(n, q);
            vec x[3], f[3], inv[3];
            for (int v = 0; v < 3; ++v) {
                x[v].resize(n); orc_fill_splitmix(x[v].data(), n, 900 + v + rnd() % 100, q);
                f[v] = x[v]; orc_ntt_fwd(f[v].data(), n, q, tb.roots.data(), tb.precon.data());
                inv[v] = x[v]; orc_ntt_inv(inv[v].data(), n, q, tb.iroots.data(), tb.iprecon.data(), tb.inv_n, tb.inv_n_w);
            }
            ws = std::min<size_t>(ws * 4, 2000);
            std::vector<vec> a(ws), c(ws);
            for (size_t b = 0; b < ws; ++b) { a[b] = x[b % 3]; c[b] = x[b % 3]; }
            _set_worksize_NTT(ws);
            for (size_t b = 0; b < ws; ++b) _NTT(a[b].data(), tb.roots.data(), tb.precon.data(), q, n);
            CHECK(_NTTCompleted(), "_NTTCompleted");
            _set_worksize_INTT(ws);
            for (size_t b = 0; b < ws; ++b) _INTT(c[b].data(), tb.iroots.data(), tb.iprecon.data(), q, tb.inv_n, tb.inv_n_w, n);
            CHECK(_INTTCompleted(), "_INTTCompleted");
            for (size_t b = 0; b < ws; ++b) {
                CHECK(a[b] == f[b % 3], "stress _NTT n=%lu bits=%u worksize %zu obj %zu", n, bits, ws, b);
                CHECK(c[b] == inv[b % 3], "stress _INTT n=%lu bits=%u worksize %zu obj %zu", n, bits, ws, b);
            }
        } else {
            const uint64_t nm = 1 + rnd() % 4;
            ws = std::min<size_t>(ws, 130);
            vec mod(nm);
            orc_generate_primes(mod.data(), nm, 30 + unsigned(rnd() % 30), n);
            vec a[3], b[3], ref[3];
            for (int v = 0; v < 3; ++v) {
                a[v].resize(2 * nm * n); b[v].resize(2 * nm * n); ref[v].resize(3 * nm * n);
                for (uint64_t m = 0; m < 2 * nm; ++m) {
                    orc_fill_splitmix(&a[v][m * n], n, 40 + v * 16 + m, mod[m % nm]);
                    orc_fill_splitmix(&b[v][m * n], n, 140 + v * 16 + m, mod[m % nm]);
                }
                orc_dyadic_multiply(ref[v].data(), a[v].data(), b[v].data(), n, mod.data(), nm, 1);
            }
            std::vector<vec> out(ws, vec(3 * nm * n));
            set_worksize_DyadicMultiply(ws);
            for (size_t k = 0; k < ws; ++k) DyadicMultiply(out[k].data(), a[k % 3].data(), b[k % 3].data(), n, mod.data(), nm);
            CHECK(DyadicMultiplyCompleted(), "DyadicMultiplyCompleted");
            for (size_t k = 0; k < ws; ++k) CHECK(out[k] == ref[k % 3], "stress dyadic n=%lu nm=%lu worksize %zu obj %zu", n, nm, ws, k);
        }
        ++windows; objects += ws;
        if (failures > 20) break;
    }
    std::printf("stress: %zu windows, %zu objects in %.0f s (seed %lu)\n", windows, objects, elapsed(), (unsigned long)seed);
}

int main(int argc, char** argv) {
    acquire_FPGA_resources();
    if (argc > 1 && !std::strcmp(argv[1], "stress")) {            // stress [seconds] [seed]
        stress(argc > 2 ? atof(argv[2]) : 30.0, argc > 3 ? strtoull(argv[3], nullptr, 10) : 1);
        release_FPGA_resources();
        std::printf(failures ? "CXX API: %d FAILURE(S)\n" : "CXX API: ALL PASSED\n", failures);
        return failures ? 1 : 0;
    }
    if (argc > 1 && !std::strcmp(argv[1], "threads")) {           // concurrent worksize-1 callers: threads [count] [rounds]
        test_concurrent_worksize1(argc > 2 ? atoi(argv[2]) : 4, argc > 3 ? atoi(argv[3]) : 20);
        release_FPGA_resources();
        std::printf(failures ? "CXX API: %d FAILURE(S)\n" : "CXX API: ALL PASSED\n", failures);
        return failures ? 1 : 0;
    }
    if (argc > 1 && !std::strcmp(argv[1], "alias")) {             // the ordering stress on its own: alias [repeats]
        test_keyswitch_aliased_across_subbatches(argc > 2 ? atoi(argv[2]) : 50);
        release_FPGA_resources();
        std::printf(failures ? "CXX API: %d FAILURE(S)\n" : "CXX API: ALL PASSED\n", failures);
        return failures ? 1 : 0;
    }
    for (unsigned bits : {20u, 32u, 55u, 62u}) test_ntt_intt(16384, bits, 4);     // test_fwd_ntt.cpp:119-170 primes
    test_ntt_intt(1024, 30, 3);
    test_dyadic(8192, 7, 4);
    test_dyadic(32768, 2, 2);
    test_keyswitch(16384, 6, 7, 3);                                               // the 16384_6_7_7_2 shape
    test_keyswitch(8192, 5, 7, 2);                                                // the 8192_5_7_6_2-like shape
    test_keyswitch(1024, 1, 2, 2);
    test_keyswitch_aliased_results();
    test_keyswitch_aliased_across_subbatches(2);
    test_mixed_and_fences();
    release_FPGA_resources();
    std::printf(failures ? "CXX API: %d FAILURE(S)\n" : "CXX API: ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
