// pmc_workload.cpp -- native workload for rocprofv3 passes (kernel trace or one PMC group per run): `reps` launches of
// hexl_keyswitch at batch B (device-resident synthetic data, words uniform below their modulus) through the C-ABI, and
// optionally `reps` forward + inverse NTT launches at batch 1024. Starts in a fraction of a second (no Python), so
// bench.py can afford to run the PMC passes of its roofline block INSIDE the benchmark run.
//   usage: pmc_workload <batch> <L> [reps = 2] [ntt = 0|1]
// Data are synthetic and nothing is checked here: parity is the test suite's job, this only feeds counters.
// With HEXL_WORKLOAD_POWER=1 the timed keyswitch loop is also sampled for board power and shader clock (amdgpu hwmon, every
// 50 ms) and one JSON line with keyswitch/s, W, MHz and mJ per keyswitch is printed: tools/byte_budget.py runs that with `reps`
// large enough for ~8 s per leg, once per stream-aliasing mask of the profiling build (pmc_workload_prof, HEXL_KSX_ALIAS).
#include <hip/hip_runtime.h>

#include <glob.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../include/hexl_mi355x.h"

#define CK(e) do { int rc_ = (int)(e); if (rc_) { std::fprintf(stderr, "%s failed: %d\n", #e, rc_); return 1; } } while (0)

// board power / shader clock of the first amdgpu hwmon that reports power (one-GPU boxes), sampled on a thread
struct PowerProbe {
    std::string dir;
    std::atomic<bool> stop{false};
    std::thread th;
    double w_sum = 0, f_sum = 0, w_max = 0;
    long n = 0, nf = 0;
    static double read(const std::string& f) {
        FILE* fp = std::fopen(f.c_str(), "r");
        if (!fp) return -1;
        double v = -1;
        if (std::fscanf(fp, "%lf", &v) != 1) v = -1;
        std::fclose(fp);
        return v;
    }
    explicit PowerProbe(const std::string& pci_bdf) {
        // the hwmon of THIS GPU (the box has eight): /sys/bus/pci/devices/<domain:bus:dev.fn>/hwmon/hwmon*
        if (!pci_bdf.empty()) {
            glob_t g{};
            if (!::glob(("/sys/bus/pci/devices/" + pci_bdf + "/hwmon/hwmon*").c_str(), 0, nullptr, &g) && g.gl_pathc) dir = g.gl_pathv[0];
            globfree(&g);
            if (!dir.empty()) return;
        }
        find_any();
    }
    void find_any() {
        // an amdgpu hwmon directory: it reports board power (power1_average or power1_input) AND the shader clock (freq1_input)
        for (const char* pat : {"/sys/class/drm/card*/device/hwmon/hwmon*", "/sys/bus/pci/devices/*/hwmon/hwmon*", "/sys/class/hwmon/hwmon*"}) {
            glob_t g{};
            if (!::glob(pat, 0, nullptr, &g))
                for (size_t i = 0; i < g.gl_pathc && dir.empty(); ++i) {
                    const std::string d = g.gl_pathv[i];
                    if (read(d + "/freq1_input") > 0 && (read(d + "/power1_average") > 0 || read(d + "/power1_input") > 0)) dir = d;
                }
            globfree(&g);
            if (!dir.empty()) break;
        }
    }
    void start() {
        if (dir.empty()) return;
        th = std::thread([this] {
            while (!stop.load()) {
                double w = read(dir + "/power1_average");
                if (w <= 0) w = read(dir + "/power1_input");
                const double f = read(dir + "/freq1_input");
                if (w > 0) { w_sum += w / 1e6; w_max = w / 1e6 > w_max ? w / 1e6 : w_max; ++n; }
                if (f > 0) { f_sum += f / 1e6; ++nf; }
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        });
    }
    void finish() { stop = true; if (th.joinable()) th.join(); }
};

static uint64_t sm_state = 7;
static uint64_t sm() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
    unsigned __int128 r = 1, b = a % q;
    for (; e; e >>= 1, b = b * b % q) if (e & 1) r = r * b % q;
    return (uint64_t)r;
}

int main(int argc, char** argv) {
    const size_t batch = argc > 1 ? (size_t)atol(argv[1]) : 256;
    const uint64_t n = 16384, L = argc > 2 ? (uint64_t)atoi(argv[2]) : 7, K = L + 1;
    const int reps = argc > 3 ? atoi(argv[3]) : 2, with_ntt = argc > 4 ? atoi(argv[4]) : 0;
    // GeneratePrimes(8, 51, 16384) of the reference's test utilities: 51-bit primes, 1 mod 2n
    const uint64_t primes[8] = {2251799814045697ull, 2251799814799361ull, 2251799814930433ull, 2251799815094273ull,
                                2251799815487489ull, 2251799815520257ull, 2251799816273921ull, 2251799816568833ull};
    // HEXL_WORKLOAD_PRIMES=top52: the LARGEST 52-bit primes = 1 mod 2n (what SEAL's CoeffModulus::Create(n, {52, ...}) picks): the strict
    // FP64 tier -- its counters beside the lazy tier's
    const uint64_t top52[8] = {4503599626682369ull, 4503599626321921ull, 4503599625830401ull, 4503599625535489ull,
                               4503599625404417ull, 4503599624847361ull, 4503599624716289ull, 4503599623864321ull};
    const bool strict = getenv("HEXL_WORKLOAD_PRIMES") && std::string(getenv("HEXL_WORKLOAD_PRIMES")) == "top52";
    if (L < 1 || K > 8 || !batch) { std::fprintf(stderr, "usage: pmc_workload <batch> <L <= 7> [reps] [ntt]\n"); return 2; }
    std::vector<uint64_t> moduli(strict ? top52 : primes, (strict ? top52 : primes) + K), msf(K, 1);
    for (uint64_t i = 0; i + 1 < K; ++i) msf[i] = powmod(moduli[K - 1] % moduli[i], moduli[i] - 2, moduli[i]);
    hexl_ctx* ctx = nullptr;
    CK(hexl_ctx_create(0, &ctx));
    hexl_ks_plan* plan = nullptr;
    CK(hexl_ks_plan_create(ctx, n, L, K, K, 2, moduli.data(), msf.data(), nullptr, &plan));
    std::vector<std::vector<uint64_t>> keys(L, std::vector<uint64_t>(2 * K * n));
    std::vector<const uint64_t*> kp;
    for (auto& k : keys) {
        for (uint64_t kk = 0; kk < 2; ++kk)
            for (uint64_t i = 0; i < K; ++i)
                for (uint64_t j = 0; j < n; ++j) k[(kk * K + i) * n + j] = sm() % moduli[i];
        kp.push_back(k.data());
    }
    CK(hexl_ks_set_keys(plan, kp.data()));
    // four distinct instances, replicated over the batch on the device
    const size_t tw = L * n, rw = 2 * L * n, distinct = batch < 4 ? batch : 4;
    std::vector<uint64_t> ht(distinct * tw), hr(distinct * rw);
    for (size_t b = 0; b < distinct; ++b) {
        for (uint64_t d = 0; d < L; ++d) for (uint64_t j = 0; j < n; ++j) ht[b * tw + d * n + j] = sm() % moduli[d];
        for (uint64_t x = 0; x < 2 * L; ++x) for (uint64_t j = 0; j < n; ++j) hr[b * rw + x * n + j] = sm() % moduli[x % L];
    }
    uint64_t *dt = nullptr, *dr = nullptr;
    CK(hipMalloc((void**)&dt, batch * tw * 8));
    CK(hipMalloc((void**)&dr, batch * rw * 8));
    CK(hipMemcpy(dt, ht.data(), distinct * tw * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, hr.data(), distinct * rw * 8, hipMemcpyHostToDevice));
    for (size_t have = distinct; have < batch; have *= 2) {
        const size_t cnt = have * 2 <= batch ? have : batch - have;
        CK(hipMemcpy(dt + have * tw, dt, cnt * tw * 8, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(dr + have * rw, dr, cnt * rw * 8, hipMemcpyDeviceToDevice));
    }
    const bool power = getenv("HEXL_WORKLOAD_POWER") && atoi(getenv("HEXL_WORKLOAD_POWER")) == 1;
    if (power) { CK(hexl_keyswitch(plan, dr, dt, batch)); CK(hexl_ctx_sync(ctx)); }      // warm-up: scratch, clocks
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, 0) != hipSuccess) bdf[0] = 0;
    for (char* c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = char(*c - 'A' + 'a');     // sysfs names are lower case
    PowerProbe probe(bdf);
    if (power) probe.start();
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hexl_keyswitch(plan, dr, dt, batch));
    CK(hexl_ctx_sync(ctx));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (power) {
        probe.finish();
        const double rate = double(batch) * reps / secs, w = probe.n ? probe.w_sum / probe.n : 0, f = probe.nf ? probe.f_sum / probe.nf : 0;
        std::printf("{\"keyswitch_per_s\": %.1f, \"seconds\": %.3f, \"batch\": %zu, \"reps\": %d, \"L\": %lu, \"board_power_w_mean\": %.1f, "
                    "\"board_power_w_max\": %.1f, \"sclk_mhz_mean\": %.1f, \"power_samples\": %ld, \"mj_per_keyswitch\": %.4f}\n",
                    rate, secs, batch, reps, (unsigned long)L, w, probe.w_max, f, probe.n, rate > 0 ? w / rate * 1e3 : 0.0);
    }
    if (with_ntt) {
        // tables need not be genuine for counters, but genuine Shoup pairs keep the exact FP64 fast path (ntt.hip) in play
        const uint64_t q = moduli[0], nb = 1024;
        std::vector<uint64_t> roots(n), precon(n);
        for (uint64_t j = 0; j < n; ++j) {
            roots[j] = sm() % q;
            precon[j] = (uint64_t)(((unsigned __int128)roots[j] << 64) / q);
        }
        uint64_t *dx = nullptr, *dro = nullptr, *dpr = nullptr;
        CK(hipMalloc((void**)&dx, nb * n * 8));
        CK(hipMalloc((void**)&dro, n * 8));
        CK(hipMalloc((void**)&dpr, n * 8));
        CK(hipMemcpy(dro, roots.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dpr, precon.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dx, dt, (nb * n <= batch * tw ? nb * n : batch * tw) * 8, hipMemcpyDeviceToDevice));
        for (int r = 0; r < reps; ++r) {
            CK(hexl_ntt_fwd(ctx, dx, nb, dro, dpr, q, n));
            CK(hexl_ntt_inv(ctx, dx, nb, dro, dpr, q, 1, 1, n));
        }
        CK(hexl_ctx_sync(ctx));
        (void)hipFree(dx); (void)hipFree(dro); (void)hipFree(dpr);
    }
    CK(hexl_ks_plan_destroy(plan));
    (void)hipFree(dt); (void)hipFree(dr);
    CK(hexl_ctx_destroy(ctx));
    std::printf("pmc_workload: %d x keyswitch batch %zu L=%lu%s done\n", reps, batch, (unsigned long)L, with_ntt ? " + NTT" : "");
    return 0;
}
