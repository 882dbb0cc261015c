// hexl-fpga.h -- public C++ interface of libhexl-fpga.so, MI355X edition.
//
// Drop-in for the reference's host/inc/hexl-fpga.h (intel/hexl-fpga v2.0, lines 15-161): the same
// fourteen free functions in namespace intel::hexl with identical signatures, so they mangle to the
// same Itanium symbols and the reference's tests/, benchmark/, examples/ and the SEAL
// bridge (which calls intel::hexl::KeySwitch through HEXL's same-signature declaration) link
// unchanged. This file is written for this repository; only the declarations are shared facts.
//
// Contract (same as the reference, host/src/fpga_int.cpp):
//   * acquire_FPGA_resources() once per process before any other call, release_FPGA_resources() at
//     the end. "FPGA" is kept in the names for compatibility: the resources are MI355X GPUs
//     (env NUM_DEV = how many, default 1).
//   * set_worksize_X(ws); exactly ws calls of X(...); XCompleted(). With ws == 1 (the default, and
//     the value after every XCompleted) X(...) completes before returning.
//   * every pointer is caller-owned and must stay valid and untouched until XCompleted() returns.
//     KeySwitch `result` is read-modify-write (the output is added into it mod q_i); _NTT/_INTT work
//     in place; DyadicMultiply writes `results` only. The k_switch_keys pointer values identify the
//     device key cache entry.
//   * no return codes and no exceptions; invalid arguments or a GPU failure print a message and
//     abort() (the reference FPGA_ASSERTs / exit()s).
#ifndef HEXL_FPGA_MI355X_PUBLIC_H
#define HEXL_FPGA_MI355X_PUBLIC_H

#include <cstdint>

namespace intel {
namespace hexl {

void acquire_FPGA_resources();
void release_FPGA_resources();

// ---- ciphertext x ciphertext dyadic multiply: (x0,x1) (*) (y0,y1) -> (x0y0, x0y1+x1y0, x1y1) per limb.
// operand[(p*n_moduli + m)*n + j], p < 2; results[(p*n_moduli + m)*n + j], p < 3; moduli[n_moduli].
void set_worksize_DyadicMultiply(uint64_t ws);
void DyadicMultiply(uint64_t* results, const uint64_t* operand1, const uint64_t* operand2, uint64_t n,
                    const uint64_t* moduli, uint64_t n_moduli);
bool DyadicMultiplyCompleted();

// ---- CKKS key switch. t_target_iter_ptr[d*n + j], d < decomp_modulus_size;
// result[(k*decomp_modulus_size + i)*n + j], k < 2 (accumulated into);
// k_switch_keys[d][(k*key_modulus_size + i)*n + j]; moduli / modswitch_factors[key_modulus_size], the
// special prime is moduli[key_modulus_size - 1]; twiddle_factors = key_modulus_size blocks of 4n words
// [inv_roots | precon_inv | roots | precon_roots] or nullptr (derived from the minimal primitive root).
void set_worksize_KeySwitch(uint64_t ws);
void KeySwitch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n, uint64_t decomp_modulus_size,
               uint64_t key_modulus_size, uint64_t rns_modulus_size, uint64_t key_component_count,
               const uint64_t* moduli, const uint64_t** k_switch_keys, const uint64_t* modswitch_factors,
               const uint64_t* twiddle_factors = nullptr);
bool KeySwitchCompleted();

// ---- standalone negacyclic NTT / inverse NTT (deprecated in the reference since v1.1, kept for its
// tests and benchmarks). Tables in bit-reversed order; inverse tables in the HEXL stage-major layout.
[[deprecated]] void _set_worksize_NTT(uint64_t ws);
[[deprecated]] void _NTT(uint64_t* operand, const uint64_t* root_of_unity_powers,
                         const uint64_t* precon_root_of_unity_powers, uint64_t coeff_modulus, uint64_t n);
[[deprecated]] bool _NTTCompleted();

[[deprecated]] void _set_worksize_INTT(uint64_t ws);
[[deprecated]] void _INTT(uint64_t* operand, const uint64_t* inv_root_of_unity_powers,
                          const uint64_t* precon_inv_root_of_unity_powers, uint64_t coeff_modulus, uint64_t inv_n,
                          uint64_t inv_n_w, uint64_t n);
[[deprecated]] bool _INTTCompleted();

}  // namespace hexl
}  // namespace intel

#endif  // HEXL_FPGA_MI355X_PUBLIC_H
