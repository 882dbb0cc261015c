/*
 * hexl_mi355x.h -- C-ABI of the MI355X (gfx950) launcher library libhexl_mi355x.so.
 *
 * This is the drop-in boundary for the hexl-fpga hot path. In the reference the same seam is
 * the dlopen'ed "bitstream" launcher table (host/inc/dl_kernel_interfaces.hpp:45-136,
 * device/fwd_ntt.cpp:619-646, device/inv_ntt.cpp:577-607, device/dyadic_multiply.cpp:349-405,
 * device/keyswitch.cpp:15-65) whose signatures carry sycl::queue& / sycl::buffer&; here it is
 * plain C: opaque handles, raw pointers and sizes, int status (0 = ok, else hipError_t or
 * a negative HEXL_E_* code). No torch / C++ types cross it.
 *
 * Pointers named d_* are DEVICE pointers (HBM), h_* are HOST pointers.
 * All launchers are asynchronous on the context's stream; hexl_ctx_sync() waits.
 *
 * The C++ API of the reference (host/inc/hexl-fpga.h:15-161, namespace intel::hexl) is built on
 * top of these entry points in libhexl-fpga.so (see include/hexl-fpga.h).
 */
#ifndef HEXL_MI355X_H
#define HEXL_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEXL_E_BADARG   (-1)   /* unsupported n / null pointer / size limit */
#define HEXL_E_NOKEYS   (-2)   /* hexl_keyswitch before hexl_ks_set_keys */
#define HEXL_E_NODEVICE (-3)   /* no gfx950 device visible */
#define HEXL_E_RANGE    (-4)   /* HEXL_KS_VALIDATE=1: a t_target / result word is not below its modulus; the call was REFUSED, nothing computed */
#define HEXL_W_RANGE    1      /* status, not an error: the call ran, but the FP64 kernels saw a t_target / result word that is not below its
                                  modulus (hexl_ks_range_check, hexl_keyswitch_host); the output words of such an object are unspecified */

typedef struct hexl_ctx hexl_ctx;         /* one per GPU: stream + scratch */
typedef struct hexl_ks_plan hexl_ks_plan; /* keyswitch parameter set: tables + keys on device */

/* number of gfx950 devices visible to the process (0 if none); what NUM_DEV is clamped to
 * (DevicePool, host/src/fpga.cpp:1646-1673) */
int hexl_device_count(void);

/* replaces acquire_/release_FPGA_resources' device half (host/src/fpga.cpp:1646-1685):
 * binds `device`, creates the stream. */
int hexl_ctx_create(int device, hexl_ctx** out);
int hexl_ctx_destroy(hexl_ctx* ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream; NULL = the legacy default stream) */
int hexl_ctx_set_stream(hexl_ctx* ctx, void* hip_stream);
/* go back to the context's own non-blocking stream (the state after hexl_ctx_create) */
int hexl_ctx_use_own_stream(hexl_ctx* ctx);
int hexl_ctx_sync(hexl_ctx* ctx);
/* library/device report for logs: writes a NUL-terminated string */
int hexl_ctx_describe(hexl_ctx* ctx, char* buf, size_t buflen);

/* K1 -- batched negacyclic forward NTT, in place, bit-exact with fwd_ntt_kernel
 * (device/fwd_ntt.cpp:82-497; launchers fwd_ntt/ntt_input/ntt_output :619-646).
 * d_x[batch][n]; one modulus and one table pair (bit-reversed order, n words each) per batch.
 * n in {1024, 2048, 4096, 8192, 16384} (reference: 16384 only, host/src/ntt.cpp:24). */
int hexl_ntt_fwd(hexl_ctx* ctx, uint64_t* d_x, size_t batch, const uint64_t* d_roots,
                 const uint64_t* d_precon, uint64_t q, uint64_t n);

/* K2 -- batched inverse NTT, in place, bit-exact with inv_ntt_kernel
 * (device/inv_ntt.cpp:83-441; launchers :577-607). Inverse tables in the HEXL layout
 * (stage-major, first used entry at index 1); inv_n, inv_n_w caller scalars
 * (host/inc/hexl-fpga.h:150-154). */
int hexl_ntt_inv(hexl_ctx* ctx, uint64_t* d_x, size_t batch, const uint64_t* d_inv_roots,
                 const uint64_t* d_inv_precon, uint64_t q, uint64_t inv_n, uint64_t inv_n_w,
                 uint64_t n);

/* K3 -- batched dyadic ciphertext multiply (device/dyadic_multiply.cpp:61-342, launchers
 * :349-405). d_a/d_b[batch][2][n_moduli][n], d_out[batch][3][n_moduli][n],
 * d_moduli[batch][n_moduli]. Exact for any 64-bit operands, moduli in [2, 2^62). */
int hexl_dyadic_multiply(hexl_ctx* ctx, uint64_t* d_out, const uint64_t* d_a,
                         const uint64_t* d_b, size_t batch, uint64_t n,
                         const uint64_t* d_moduli, uint64_t n_moduli);

/* K4 -- keyswitch. The plan replaces the reference's per-parameter-set device state:
 * build_modulus_meta / build_invn_meta / KeySwitch_load_twiddles (host/src/fpga.cpp:1049-1123)
 * and KeySwitch_load_keys (:1167-1248).
 *   n in {1024..16384}; 1 <= L < K <= 16; key_component_count == 2; moduli < 2^60
 *   (reference: K <= 7, moduli <= 2^52, host/src/keyswitch.cpp:23-34).
 *   h_moduli[K], h_modswitch[K]; h_twiddles = K blocks of 4n words
 *   [inv_roots | precon_inv | roots | precon_roots] in the hexl-fpga layout
 *   (host/src/twiddle-factors.cpp:16-62) or NULL to derive them from
 *   MinimalPrimitiveRoot(2n, q_i) as fpga.cpp:1097-1109 does. */
int hexl_ks_plan_create(hexl_ctx* ctx, uint64_t n, uint64_t decomp_modulus_size,
                        uint64_t key_modulus_size, uint64_t rns_modulus_size,
                        uint64_t key_component_count, const uint64_t* h_moduli,
                        const uint64_t* h_modswitch, const uint64_t* h_twiddles,
                        hexl_ks_plan** out);
int hexl_ks_plan_destroy(hexl_ks_plan* plan);
/* h_keys[d] -> key words k_switch_keys[d][(k*K + i)*n + j] (fpga.cpp:1186-1190), d < L */
int hexl_ks_set_keys(hexl_ks_plan* plan, const uint64_t* const* h_keys);
/* d_t_target[batch][L][n]; d_result[batch][2][L][n] is read-modify-write: the keyswitch
 * output is added into it mod q_i (fpga.cpp:441-475). Precondition, as for intel::hexl::KeySwitch:
 * every t_target / result word is below its modulus (the FP64 kernels used for moduli < 2^52 compute
 * the exact residues of in-range data; the integer kernels used for larger moduli replay the lazy
 * arithmetic on raw words instead -- out-of-range inputs are outside the contract on both).
 * With HEXL_KS_VALIDATE=1 in the environment every call first checks that precondition on the device
 * (one extra pass over the inputs and a stream synchronisation) and returns HEXL_E_RANGE without
 * touching `result` if it does not hold. Steps load -> INTT -> mod-up -> NTT ->
 * key MAC -> INTT(special) -> round -> NTT -> mod-switch -> store of
 * device/keyswitch/ (SURVEY 2.1-K4). */
int hexl_keyswitch(hexl_ks_plan* plan, uint64_t* d_result, const uint64_t* d_t_target,
                   size_t batch);
/* The FP64 kernels (moduli < 2^52) check the precondition above where they convert the words anyway -- one compare per
 * word, no extra pass -- and OR the outcome into a flag of the plan. This call waits for the plan's stream and returns
 * HEXL_W_RANGE (> 0) if any hexl_keyswitch launched on the plan since the flag was last cleared saw a t_target / result word
 * >= its modulus (the output words of that instance are then unspecified), 0 otherwise; it clears the flag. The integer
 * kernels (moduli >= 2^52) do not flag: they replay the reference's lazy arithmetic on whatever words they get.
 * hexl_keyswitch_host() reports for ITS OWN objects only: it clears the flag when it starts (call hexl_ks_range_check first
 * if earlier device-pointer launches on the plan matter) and returns HEXL_W_RANGE when one or more objects of the call had
 * an out-of-range word -- it cannot say which. HEXL_E_RANGE (< 0) is different: the HEXL_KS_VALIDATE=1 refusal, nothing ran. */
int hexl_ks_range_check(hexl_ks_plan* plan);
/* Beyond the reference's envelope (the use-case of its combined image,
 * device/dyadic_multiply_keyswitch.cpp:4-5): ciphertext multiply + relinearize in one pass.
 *   d_a, d_b [batch][2][L][n] (the DyadicMultiply operand layout with n_moduli = L, words < q_i);
 *   d_out    [batch][2][L][n] is WRITTEN with (a0 b0, a0 b1 + a1 b0) + KeySwitch(a1 b1), i.e. what
 *   DyadicMultiply followed by KeySwitch(result = components 0..1, t_target = component 2) leaves -- but the
 *   three-component product never exists in memory. n = 1024 ... 32768 and moduli < 2^52 only (else HEXL_E_BADARG).
 *   d_out must not overlap d_a or d_b (component 0 is stored before component 1's operands are read): HEXL_E_BADARG. */
int hexl_multiply_relinearize(hexl_ks_plan* plan, uint64_t* d_out, const uint64_t* d_a,
                              const uint64_t* d_b, size_t batch);
/* Arithmetic tier per limb (introspection for logs and tests): tiers[i], i < key_modulus_size, = the forward transforms' range-
 * reduction period modulo q_i on the FP64 path -- 12 / 6 / 3 for q_i <= 2^49 / 2^50 / 2^51 (1 + 2^-7), 0 = every value reduced after
 * every operation (q_i up to 2^52); -1 for every limb of a plan on the integer kernels (a modulus >= 2^52). Every transform runs modulo
 * ONE q_i and takes that limb's tier, as each NTT engine of the reference runs on its own modulus (device/keyswitch/ntt_core.hpp:285-291);
 * HEXL_KS_PER_LIMB=0 gives every limb the tier of the plan's largest modulus. Returns 1 when the limbs in use differ in tier, else 0. */
int hexl_ks_plan_tiers(const hexl_ks_plan* plan, int* tiers /* [key_modulus_size] */);
/* bytes of HBM scratch a batch of `batch` keyswitches needs (for capacity planning) */
size_t hexl_ks_scratch_bytes(const hexl_ks_plan* plan, size_t batch);

/* Host-pointer entry points used by the C++ API layer (libhexl-fpga.so): pinned staging + H2D/D2H
 * around the launchers above; synchronous on return. Every object is passed by its own pointer -- the
 * reference needs batch elements contiguous for NTT/INTT/dyadic (one memcpy from the first object,
 * host/src/fpga.cpp:379-388,405-406) and copies keyswitch objects one by one (fpga.cpp:542-555); per-object
 * pointers are a superset of both. */
int hexl_ntt_fwd_host(hexl_ctx* ctx, uint64_t* const* h_x, size_t batch, const uint64_t* h_roots,
                      const uint64_t* h_precon, uint64_t q, uint64_t n);
int hexl_ntt_inv_host(hexl_ctx* ctx, uint64_t* const* h_x, size_t batch, const uint64_t* h_inv_roots,
                      const uint64_t* h_inv_precon, uint64_t q, uint64_t inv_n,
                      uint64_t inv_n_w, uint64_t n);
int hexl_dyadic_multiply_host(hexl_ctx* ctx, uint64_t* const* h_out, const uint64_t* const* h_a,
                              const uint64_t* const* h_b, size_t batch, uint64_t n,
                              const uint64_t* const* h_moduli, uint64_t n_moduli);
int hexl_keyswitch_host(hexl_ks_plan* plan, uint64_t* const* h_results,
                        const uint64_t* const* h_t_targets, size_t batch);

/* timing hook for bench.py: average milliseconds per keyswitch launch and per stage (s1: steps 1-2, inverse
 * transforms + mod-up transforms; s2: steps 3-4, multiply-accumulate + special-prime inverse; s3: steps 5-7,
 * mod-down) over `iters` launches of a batch that fits one scratch chunk, measured with hipEvents on the
 * context's stream. */
int hexl_ks_time_stages(hexl_ks_plan* plan, uint64_t* d_result, const uint64_t* d_t_target,
                        size_t batch, int iters, float* ms_out /* [4]: total, s1, s2, s3 */);

#ifdef __cplusplus
}
#endif
#endif /* HEXL_MI355X_H */
