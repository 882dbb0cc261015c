// ref_shim.cpp -- thin extern "C" window onto the REFERENCE's own CPU code, compiled from the
// sources where they lie under /root/reference (never copied into this repo):
//   tests/test_utils/ntt.cpp            the reference tests' CPU NTT oracle (what its gtests
//                                        ASSERT_EQ the FPGA output against)
//   host/src/number_theory_util.cpp,
//   host/src/twiddle-factors.cpp        host number theory + keyswitch twiddle layout
// Output goes to oracle/_ref/libhexlfpga_ref.so (git-ignored). TEST INFRASTRUCTURE ONLY:
// used to validate oracle/hexl_oracle.c and to regenerate tests/golden/*.json.
#include <cstdint>
#include <cstring>
#include <vector>

#include "ntt.hpp"                 // /root/reference/tests/test_utils
#include "number_theory_util.h"    // /root/reference/host/inc

extern "C" {

// NTT::NTTImpl tables (HEXL layout) for degree n, modulus q, minimal primitive root.
uint64_t ref_ntt_tables(uint64_t n, uint64_t q, uint64_t* roots, uint64_t* precon,
                        uint64_t* inv_roots, uint64_t* inv_precon) {
    hetest::utils::NTT ntt(n, q);
    auto& impl = *ntt.m_impl;
    std::memcpy(roots, impl.GetRootOfUnityPowersPtr(), n * 8);
    std::memcpy(precon, impl.GetPrecon64RootOfUnityPowersPtr(), n * 8);
    std::memcpy(inv_roots, impl.GetInvRootOfUnityPowersPtr(), n * 8);
    std::memcpy(inv_precon, impl.GetPrecon64InvRootOfUnityPowersPtr(), n * 8);
    return impl.GetMinimalRootOfUnity();
}

// exactly what tests/test_fwd_ntt.cpp:103-115 / test_inv_ntt.cpp compute as "expected"
void ref_ntt_forward(uint64_t* out, const uint64_t* in, uint64_t n, uint64_t q) {
    hetest::utils::NTT ntt(n, q);
    ntt.m_impl->ComputeForward(out, in, 1, 1);
}
void ref_ntt_inverse(uint64_t* out, const uint64_t* in, uint64_t n, uint64_t q) {
    hetest::utils::NTT ntt(n, q);
    ntt.m_impl->ComputeInverse(out, in, 1, 1);
}
// free-function forms taking caller tables (bench-style random tables allowed)
void ref_fwd_with_tables(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* roots,
                         const uint64_t* precon) {
    hetest::utils::ForwardTransformToBitReverse64(x, n, q, roots, precon, 1, 1);
}
void ref_inv_with_tables(uint64_t* x, uint64_t n, uint64_t q, const uint64_t* inv_roots,
                         const uint64_t* inv_precon) {
    hetest::utils::InverseTransformFromBitReverse64(x, n, q, inv_roots, inv_precon, 1, 1);
}

size_t ref_generate_primes(uint64_t* out, size_t num, size_t bits, size_t ntt_size) {
    std::vector<uint64_t> p = hetest::utils::GeneratePrimes(num, bits, ntt_size);
    std::memcpy(out, p.data(), p.size() * 8);
    return p.size();
}
uint64_t ref_minimal_primitive_root(uint64_t degree, uint64_t q) {
    return intel::hexl::fpga::MinimalPrimitiveRoot(degree, q);
}
uint64_t ref_inverse_mod(uint64_t a, uint64_t q) { return intel::hexl::fpga::InverseUIntMod(a, q); }

// hexl-fpga keyswitch twiddle block [inv | precon_inv | roots | precon_roots], 4n words
void ref_ks_tables(uint64_t n, uint64_t q, uint64_t* block4n) {
    uint64_t w = intel::hexl::fpga::MinimalPrimitiveRoot(2 * n, q);
    uint64_t bits = 0;
    while ((1ULL << bits) < n) bits++;
    intel::hexl::fpga::ComputeRootOfUnityPowers(q, n, bits, w, block4n, block4n + n,
                                                block4n + 2 * n, block4n + 3 * n);
}
}
