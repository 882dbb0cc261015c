// bench_cxx_api.cpp -- end-to-end (host pointers, PCIe included) throughput of the reference's public C++ API on
// libhexl-fpga.so, shaped like benchmark/bench_keyswitch.cpp:113-131 and bench_fwd_ntt.cpp:46-62: one warm-up
// window, then timed worksize windows. Synthetic in-range data (splitmix). Prints keyswitch/s and NTT/s.
//   usage: bench_cxx_api [worksize = 256] [L = 6] [ntt = 1|0] [json = 0|1] [chain = 52bit|seal]
// json = 1 prints one JSON object instead (bench.py's `extra.cxx_api_end_to_end` leg: SURVEY 8d's end-to-end measurement at the
// batch sizes of benchmark/micro_keyswitch.sh -- these rates include PCIe and the host copies and are never `value`).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/hexl-fpga.h"
#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
using namespace intel::hexl;
typedef std::vector<uint64_t> vec;

static uint64_t sm_state = 1;
static uint64_t sm() { uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t batch = argc > 1 ? atoi(argv[1]) : 256;
    const bool with_ntt = argc > 3 ? atoi(argv[3]) != 0 : true, json = argc > 4 && atoi(argv[4]) == 1;
    // chain = "seal": bridge-seal's prime chain (experimental/bridge-seal/tests/seal_test.sh:20: 52,30,30,40,27,27,27 bits -- the largest
    // primes = 1 mod 2n of each size, as SEAL's CoeffModulus::Create picks them), 7 key moduli whatever L: L = 6 is its relinearisation,
    // L = 5 the rotation behind the rescale
    const bool seal = argc > 5 && std::string(argv[5]) == "seal";
    const uint64_t n = 16384, L = argc > 2 ? atoi(argv[2]) : 6, K = seal ? 7 : L + 1;
    // the eight 52-bit primes of SURVEY 8c (GeneratePrimes(8, 51, 16384))
    const uint64_t primes[8] = {2251799814045697ull, 2251799814799361ull, 2251799814930433ull, 2251799815094273ull,
                                2251799815487489ull, 2251799815520257ull, 2251799816273921ull, 2251799816568833ull};
    const uint64_t seal_primes[7] = {4503599626682369ull, 1073643521ull, 1073479681ull, 1099510054913ull, 133857281ull, 132710401ull, 132612097ull};
    vec moduli(seal ? seal_primes : primes, (seal ? seal_primes : primes) + K), msf(K, 12345);
    std::vector<vec> keys(L, vec(2 * K * n));
    for (auto& k : keys) for (uint64_t kk = 0; kk < 2; ++kk) for (uint64_t i = 0; i < K; ++i) for (uint64_t j = 0; j < n; ++j) k[(kk * K + i) * n + j] = sm() % moduli[i];
    std::vector<const uint64_t*> kp; for (auto& k : keys) kp.push_back(k.data());
    std::vector<vec> t(batch, vec(L * n)), r(batch, vec(2 * L * n));
    for (size_t b = 0; b < batch; ++b) {
        for (uint64_t d = 0; d < L; ++d) for (uint64_t j = 0; j < n; ++j) t[b][d * n + j] = sm() % moduli[d];
        for (uint64_t x = 0; x < 2 * L; ++x) for (uint64_t j = 0; j < n; ++j) r[b][x * n + j] = sm() % moduli[x % L];
    }
    acquire_FPGA_resources();
    auto window = [&]() {
        set_worksize_KeySwitch(batch);
        for (size_t b = 0; b < batch; ++b) KeySwitch(r[b].data(), t[b].data(), n, L, K, K, 2, moduli.data(), kp.data(), msf.data());
        KeySwitchCompleted();
    };
    window();
    // at least 5 windows and at least ~0.3 s: a lone keyswitch takes a quarter of a millisecond
    int iters = 5;
    double t0 = now();
    for (int i = 0; i < iters; ++i) window();
    double dt = now() - t0;
    if (dt < 0.3) {
        iters = (int)(0.3 / (dt / iters)) + 1;
        t0 = now();
        for (int i = 0; i < iters; ++i) window();
        dt = now() - t0;
    }
    if (json)
        std::printf("{\"worksize\": %zu, \"L\": %lu, \"keyswitch_per_s\": %.1f, \"ms_per_window\": %.4f, \"pcie_GBps\": %.2f, \"windows\": %d",
                    batch, L, batch * iters / dt, dt / iters * 1e3, batch * iters * 3.0 * L * n * 8 / dt / 1e9, iters);
    else
    std::printf("C++ API end-to-end keyswitch N=%lu L=%lu K=%lu batch=%zu: %.0f keyswitch/s (%.2f ms/window, %.2f GB/s over PCIe)\n",
                n, L, K, batch, batch * iters / dt, dt / iters * 1e3, batch * iters * 3.0 * L * n * 8 / dt / 1e9);
    if (!with_ntt) {
        if (json) std::printf("}\n");
        release_FPGA_resources();
        return 0;
    }
    // NTT: 4096 polynomials, one modulus (bench_fwd_ntt.cpp shape), proper tables not needed for timing
    const size_t ws = 4096;
    vec x(ws * n), roots(n), precon(n);
    for (auto& v : x) v = sm() % moduli[0];
    for (uint64_t j = 0; j < n; ++j) { roots[j] = sm() % moduli[0]; precon[j] = sm(); }
    auto ntt_window = [&]() {
        _set_worksize_NTT(ws);
        for (size_t b = 0; b < ws; ++b) _NTT(x.data() + b * n, roots.data(), precon.data(), moduli[0], n);
        _NTTCompleted();
    };
    ntt_window();
    t0 = now();
    for (int i = 0; i < 3; ++i) ntt_window();
    dt = now() - t0;
    if (json)
        std::printf(", \"ntt_worksize\": %zu, \"fwd_ntt_per_s\": %.1f, \"ntt_pcie_GBps\": %.2f}\n", ws, ws * 3 / dt, ws * 3 * 2.0 * n * 8 / dt / 1e9);
    else
    std::printf("C++ API end-to-end fwd NTT N=%lu ws=%zu: %.0f NTT/s (%.2f ms/window, %.2f GB/s over PCIe)\n", n, ws, ws * 3 / dt,
                dt / 3 * 1e3, ws * 3 * 2.0 * n * 8 / dt / 1e9);
    release_FPGA_resources();
    return 0;
}
