// dyadic.hip -- K3: batched ciphertext x ciphertext dyadic multiply for gfx950.
// Replaces device/dyadic_multiply.cpp (input_fifo_kernel :61-121, operands_fetcher :231-270,
// dyadic_multiply_eu_kernel :195-228, dyadic_multiply_kernel :273-342, output_nb_fifo :124-192)
// and MultMod/AddMod of device/mod_ops.hpp:21-84.
//
// Pure streaming op: 56 bytes of HBM traffic per coefficient-limb (2x2 words in, 3 out), 4 modmuls +
// 1 modadd. One thread owns two adjacent coefficients of one (item, limb) row so every access is a
// 16-byte vector load/store and a wave touches 1 KiB contiguous per stream.
#include "hexl_internal.hpp"
#include "ntt_core.hpp"

using namespace hx;

struct DyMeta {   // per (item, limb)
    u64 q;        // modulus
    u64 mu;       // floor(2^64/q): operand reduction
    u64 len;      // floor(log2 q) - 1         (host/src/fpga.cpp:366-373)
    u64 barr_lo;  // floor(2^(len+64)/q)
};

// floor(2^(s+64)/q) truncated to 64 bits by long division (one thread per modulus, runs once per call)
__device__ static u64 div_pow2(u32 s, u64 q) {
    // numerator = 2^(s+64): feed bits from the top; remainder < q < 2^63
    u64 rem = 0, quo = 0;
    for (int bit = int(s) + 64; bit >= 0; --bit) {
        rem = (rem << 1) | (bit == int(s) + 64 ? 1 : 0);
        quo <<= 1;
        if (rem >= q) { rem -= q; quo |= 1; }
    }
    return quo;
}

__global__ void k_dyadic_meta(DyMeta* meta, const u64* __restrict__ moduli, u32 count) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 q = moduli[i];
    DyMeta m;
    m.q = q;
    const u32 fl = 63 - __clzll((long long)q);      // floor(log2 q), q >= 2
    m.len = fl - 1;                                  // q in {2,3}: len = 0
    m.barr_lo = div_pow2(u32(m.len), q);
    m.mu = div_pow2(0, q);                           // q >= 2 so it fits 64 bits
    meta[i] = m;
}

// exact x mod q for any 64-bit x (the reference only conditionally subtracts 2q and q,
// mod_ops.hpp:34-47, i.e. assumes x < 4q; tests feed operands >> 4q with toy moduli and expect
// the mathematical result, tests/test_dyadic_multiply.cpp:35-82)
__device__ __forceinline__ u64 reduce_operand(u64 x, u64 q, u64 mu) {
    u64 r = x - mulhi(x, mu) * q;
    return csub(r, q);
}

// mod_ops.hpp:49-83 with both operands already < q. The quotient estimate c3 falls short of floor(x y / q) by less than
// 1/2 + x y / 2^(k + 62) + 1 (k = bit length of q: the truncation of c1, of barr_lo, the final floor) -- at most ONE for q < 2^61, the
// range HEXL's EltwiseMultMod documents and the reference's single conditional subtraction (:82) covers, but up to TWO for q in
// [2^61, 2^62): there x y mod q came back as a value in [q, 2q) for about one operand pair in 10^4 (tools/soak_dyadic_random.py,
// round 6; the reference's own kernel does the same). The second subtraction makes the result the mathematical one -- what the
// reference's test model computes (tests/test_dyadic_multiply.cpp:59-82) -- for every modulus in [2, 2^62). (lo - c3 q < 3 q < 2^64.)
__device__ __forceinline__ u64 mulmod_barrett(u64 x, u64 y, u64 q, u32 len, u64 barr_lo) {
    const u64 lo = x * y, hi = mulhi(x, y);
    const u64 c1 = len ? ((lo >> len) | (hi << (64 - len))) : lo;
    const u64 c3 = mulhi(c1, barr_lo);
    return csub(csub(lo - c3 * q, q), q);
}

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// VEC = 2: two adjacent coefficients per thread (n even; 16-byte accesses). VEC = 1: any n (the reference only
// requires n > 0, host/src/dyadic_multiply.cpp:15-26, and its tests go down to n = 256).
template <int VEC>
__global__ __launch_bounds__(256) void k_dyadic(u64* __restrict__ out, const u64* __restrict__ a,
                                                const u64* __restrict__ b,
                                                const DyMeta* __restrict__ meta, u32 n, u32 n_moduli,
                                                u32 pairs_per_row /* n/VEC */, u64 total_pairs) {
    const u64 gid = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= total_pairs) return;
    const u64 row = gid / pairs_per_row;              // (item, limb)
    const u32 j = u32(gid - row * pairs_per_row) * VEC;
    const u64 item = row / n_moduli;
    const u32 m = u32(row - item * n_moduli);
    const DyMeta md = meta[row];
    const u64 q = md.q;
    const u32 len = u32(md.len);

    const u64 in_base = (item * 2 * n_moduli + m) * u64(n) + j;        // x0 / y0
    const u64 in_p1 = in_base + u64(n_moduli) * n;                     // x1 / y1
    u64x2 x0, x1, y0, y1;
    if constexpr (VEC == 2) {
        x0 = *reinterpret_cast<const u64x2*>(a + in_base);
        x1 = *reinterpret_cast<const u64x2*>(a + in_p1);
        y0 = *reinterpret_cast<const u64x2*>(b + in_base);
        y1 = *reinterpret_cast<const u64x2*>(b + in_p1);
    } else {
        x0[0] = a[in_base]; x1[0] = a[in_p1]; y0[0] = b[in_base]; y1[0] = b[in_p1];
        x0[1] = x1[1] = y0[1] = y1[1] = 0;
    }

    u64x2 r0, r1, r2;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        u64 a0 = x0[e], a1 = x1[e], b0 = y0[e], b1 = y1[e];
        // in-range data (the benchmark / SEAL case) skips the operand reduction wave-uniformly
        if (__any((a0 >= q) | (a1 >= q) | (b0 >= q) | (b1 >= q))) {
            a0 = reduce_operand(a0, q, md.mu);
            a1 = reduce_operand(a1, q, md.mu);
            b0 = reduce_operand(b0, q, md.mu);
            b1 = reduce_operand(b1, q, md.mu);
        }
        r0[e] = mulmod_barrett(a0, b0, q, len, md.barr_lo);                       // x0*y0
        const u64 t0 = mulmod_barrett(a0, b1, q, len, md.barr_lo);
        const u64 t1 = mulmod_barrett(a1, b0, q, len, md.barr_lo);
        r1[e] = csub(t0 + t1, q);                                                 // AddMod, mod_ops.hpp:21-29
        r2[e] = mulmod_barrett(a1, b1, q, len, md.barr_lo);                       // x1*y1
    }
    const u64 out_base = (item * 3 * n_moduli + m) * u64(n) + j;
    const u64 stride = u64(n_moduli) * n;
    if constexpr (VEC == 2) {
        *reinterpret_cast<u64x2*>(out + out_base) = r0;
        *reinterpret_cast<u64x2*>(out + out_base + stride) = r1;
        *reinterpret_cast<u64x2*>(out + out_base + 2 * stride) = r2;
    } else {
        out[out_base] = r0[0]; out[out_base + stride] = r1[0]; out[out_base + 2 * stride] = r2[0];
    }
}

int hx_launch_dyadic(hexl_ctx* ctx, u64* d_out, const u64* d_a, const u64* d_b, size_t batch, u64 n,
                     const u64* d_moduli, u64 n_moduli) {
    if (!batch || !n_moduli) return 0;
    if (n == 0 || n > (1u << 24)) return HEXL_E_BADARG;
    const size_t rows = batch * n_moduli;
    int rc = hx_reserve_device(ctx, &ctx->d_meta, &ctx->d_meta_bytes, rows * sizeof(DyMeta));
    if (rc) return rc;
    DyMeta* meta = (DyMeta*)ctx->d_meta;
    hipLaunchKernelGGL(k_dyadic_meta, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, meta,
                       d_moduli, (u32)rows);
    // 16-byte accesses need every row to start 16-byte aligned: n even and 16-byte aligned base pointers
    const bool vec2 = (n % 2 == 0) && (((uintptr_t)d_out | (uintptr_t)d_a | (uintptr_t)d_b) % 16 == 0);
    if (vec2) {
        const u64 total = u64(rows) * (n / 2);
        hipLaunchKernelGGL(k_dyadic<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_out, d_a,
                           d_b, meta, (u32)n, (u32)n_moduli, (u32)(n / 2), total);
    } else {
        const u64 total = u64(rows) * n;
        hipLaunchKernelGGL(k_dyadic<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_out, d_a,
                           d_b, meta, (u32)n, (u32)n_moduli, (u32)n, total);
    }
    return (int)hipGetLastError();
}
