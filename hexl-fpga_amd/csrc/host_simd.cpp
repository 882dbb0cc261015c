// host_simd.cpp -- the HOST half of the reference's KeySwitch contract: `result += output` per limb
// (FPGAObject_KeySwitch::fill_out_data, host/src/fpga.cpp:441-475: the host adds what the device produced into the caller's
// array). On the host-pointer path this accumulate is the slowest stage of the staging pipeline (round 6 trace at worksize 128:
// 1.15-1.5 ms per 32 objects against 0.9 ms for their PCIe download), and the scalar loop clang / gcc emit for the baseline x86-64 ISA
// (no unsigned 64-bit vector compare before AVX-512, no 64-bit vector compare at all in SSE2) is what held it there -- so it is written
// with AVX-512 / AVX2 intrinsics behind a run-time CPU check; any other host takes the portable loop.
// Plain C++ (no HIP): built by g++ / clang++ into libhexl_mi355x.so, and into the CPU staging model of tests/cpp as it is.
#include <stddef.h>
#include <stdint.h>

#if defined(__x86_64__)
#include <immintrin.h>

__attribute__((target("avx512f"))) static void add_mod_avx512(uint64_t* r, const uint64_t* o, size_t n, uint64_t q) {
    const __m512i vq = _mm512_set1_epi64((long long)q);
    size_t j = 0;
    for (; j + 16 <= n; j += 16) {
        const __m512i a0 = _mm512_loadu_si512(r + j), a1 = _mm512_loadu_si512(r + j + 8);
        const __m512i v0 = _mm512_add_epi64(a0, _mm512_loadu_si512(o + j)), v1 = _mm512_add_epi64(a1, _mm512_loadu_si512(o + j + 8));
        _mm512_storeu_si512(r + j, _mm512_mask_sub_epi64(v0, _mm512_cmpge_epu64_mask(v0, vq), v0, vq));
        _mm512_storeu_si512(r + j + 8, _mm512_mask_sub_epi64(v1, _mm512_cmpge_epu64_mask(v1, vq), v1, vq));
    }
    for (; j < n; ++j) { const uint64_t v = r[j] + o[j]; r[j] = v - (q & (0 - (uint64_t)(v >= q))); }
}

// AVX2 has signed 64-bit compares only: flip the sign bits (v >= q unsigned  <=>  (v ^ 2^63) > ((q - 1) ^ 2^63) signed; q >= 1)
__attribute__((target("avx2"))) static void add_mod_avx2(uint64_t* r, const uint64_t* o, size_t n, uint64_t q) {
    const __m256i vq = _mm256_set1_epi64x((long long)q), sign = _mm256_set1_epi64x((long long)0x8000000000000000ull);
    const __m256i lim = _mm256_xor_si256(_mm256_set1_epi64x((long long)(q - 1)), sign);
    size_t j = 0;
    for (; j + 4 <= n; j += 4) {
        const __m256i v = _mm256_add_epi64(_mm256_loadu_si256((const __m256i*)(r + j)), _mm256_loadu_si256((const __m256i*)(o + j)));
        const __m256i ge = _mm256_cmpgt_epi64(_mm256_xor_si256(v, sign), lim);
        _mm256_storeu_si256((__m256i*)(r + j), _mm256_sub_epi64(v, _mm256_and_si256(ge, vq)));
    }
    for (; j < n; ++j) { const uint64_t v = r[j] + o[j]; r[j] = v - (q & (0 - (uint64_t)(v >= q))); }
}
#endif

static void add_mod_portable(uint64_t* r, const uint64_t* o, size_t n, uint64_t q) {
    for (size_t j = 0; j < n; ++j) { const uint64_t v = r[j] + o[j]; r[j] = v - (q & (0 - (uint64_t)(v >= q))); }
}

// r[j] = r[j] + o[j] - (q if the sum reached q), j < n: words below q < 2^63 stay below q (fpga.cpp:453-468 on canonical words). The words of
// both arrays are taken as they are -- a sum that wraps 2^64 is outside every caller's contract, as it is in the reference.
void hx_add_mod_u64(uint64_t* r, const uint64_t* o, size_t n, uint64_t q) {
#if defined(__x86_64__)
    static const int level = __builtin_cpu_supports("avx512f") ? 2 : __builtin_cpu_supports("avx2") ? 1 : 0;
    if (level == 2) return add_mod_avx512(r, o, n, q);
    if (level == 1) return add_mod_avx2(r, o, n, q);
#endif
    add_mod_portable(r, o, n, q);
}

// which of the three the host runs (0 portable, 1 AVX2, 2 AVX-512): tests compare all available ones against the portable loop
int hx_add_mod_level(void) {
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx512f") ? 2 : __builtin_cpu_supports("avx2") ? 1 : 0;
#else
    return 0;
#endif
}
void hx_add_mod_u64_at_level(uint64_t* r, const uint64_t* o, size_t n, uint64_t q, int level) {
#if defined(__x86_64__)
    if (level == 2 && __builtin_cpu_supports("avx512f")) return add_mod_avx512(r, o, n, q);
    if (level == 1 && __builtin_cpu_supports("avx2")) return add_mod_avx2(r, o, n, q);
#endif
    add_mod_portable(r, o, n, q);
}

// ---- NUMA placement of the library's OWN host threads (the copy pool and the unpack lane of capi.hip) ---------------------------------
// Round 6 measured the host-pointer KeySwitch at worksize 128 at 19.5-20.0 k keyswitch/s with the process free to roam both sockets and
// 22.7-23.2 k bound to the socket the GPU hangs off (profiles/r06_host_numa_sweep.txt): the pinned staging slabs live on the GPU's node,
// and copy threads on the other socket pull every byte across the inter-socket links. The library never touches the affinity of a
// caller's thread; its own workers ask for the CPUs of the device's node (HEXL_HOST_PIN=0 leaves them alone).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__linux__)
#include <sched.h>
#endif

// NUMA node of a PCI device ("0000:dc:00.0", lower case), -1 when sysfs does not say
int hx_numa_node_of_pci(const char* bdf) {
    if (!bdf || !*bdf) return -1;
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// restrict the CALLING thread to the CPUs of `node` (intersected with what the process may use); false = nothing changed
bool hx_pin_this_thread_to_node(int node) {
#if defined(__linux__)
    if (node < 0) return false;
    char path[128];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int count = 0;
    for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {   // "0-63,128-191"
        int lo = 0, hi = 0;
        const int k = sscanf(tok, "%d-%d", &lo, &hi);
        if (k < 1) continue;
        if (k == 1) hi = lo;
        for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); ++count; }
    }
    if (!count) return false;
    return sched_setaffinity(0, sizeof want, &want) == 0;
#else
    (void)node;
    return false;
#endif
}
