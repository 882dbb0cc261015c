// ntt.hip -- standalone batched forward / inverse negacyclic NTT kernels (K1, K2) for gfx950.
// Replaces device/fwd_ntt.cpp (fwd_ntt_kernel :82-497, ntt_input_kernel :499-580,
// ntt_output_kernel :582-604) and device/inv_ntt.cpp (:83-571) of the reference.
// One workgroup per polynomial; see ntt_core.hpp for the register/LDS mapping.
#include <stdlib.h>

#include "hexl_internal.hpp"
#include "ntt_core.hpp"

using namespace hx;

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_fwd(u64* __restrict__ x,
                                                                const u64* __restrict__ roots,
                                                                const u64* __restrict__ precon, u64 q,
                                                                u32 batch) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    {
        const u32 p = blockIdx.x;   // one workgroup per polynomial (grid == batch)
        if (p >= batch) return;
        u64* px = x + size_t(p) * G::N;
        u64 v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = px[G::idxA(r, tid)];
        WgNtt<LOGN, LOGE>::forward_lazy(v, lds, tid, roots, precon, q);
        WgNtt<LOGN, LOGE>::final_reduce(v, q);
#pragma unroll
        for (int r = 0; r < G::E; ++r) px[G::idxB(r, tid)] = v[r];
    }
}

template <int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void k_ntt_inv(u64* __restrict__ x,
                                                                const u64* __restrict__ iroots,
                                                                const u64* __restrict__ iprecon, u64 q,
                                                                u64 inv_n, u64 inv_n_p, u64 inv_n_w,
                                                                u64 inv_n_w_p, u32 batch) {
    using G = Geom<LOGN, LOGE>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    {
        const u32 p = blockIdx.x;   // one workgroup per polynomial (grid == batch)
        if (p >= batch) return;
        u64* px = x + size_t(p) * G::N;
        u64 v[G::E];
#pragma unroll
        for (int r = 0; r < G::E; ++r) v[r] = px[G::idxB(r, tid)];
        WgNtt<LOGN, LOGE>::inverse(v, lds, tid, iroots, iprecon, q, inv_n, inv_n_p, inv_n_w, inv_n_w_p);
#pragma unroll
        for (int r = 0; r < G::E; ++r) px[G::idxA(r, tid)] = v[r];
    }
}

u32 hx_loge_for(u32 logn) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("HEXL_NTT_LOGE");
        forced = e ? atoi(e) : 0;
    }
    // N = 16384: 16 coefficients per thread x 1024 threads (4 waves/SIMD) measured ~10 % faster than 32 x 512
    if (logn == 14) return forced == 5 ? 5 : 4;
    return logn <= 10 ? 4 : 5;
}

u32 hx_idxB(u32 logn, u32 r, u32 tid) {
    const u32 loge = hx_loge_for(logn);
    const u32 P = (logn + loge - 1) / loge, KL = logn - (P - 1) * loge;
    return ((r >> KL) << (logn - loge + KL)) + (tid << KL) + (r & ((1u << KL) - 1));
}

template <int LOGN, int LOGE>
static int launch_fwd(hexl_ctx* ctx, u64* x, size_t batch, const u64* roots, const u64* precon, u64 q) {
    using G = Geom<LOGN, LOGE>;
    static bool attr_set = false;
    if (!attr_set) {
        HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_fwd<LOGN, LOGE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_ntt_fwd<LOGN, LOGE>), dim3((unsigned)batch), dim3(G::T), G::LDS_BYTES, ctx->stream, x,
                       roots, precon, q, (u32)batch);
    return (int)hipGetLastError();
}

template <int LOGN, int LOGE>
static int launch_inv(hexl_ctx* ctx, u64* x, size_t batch, const u64* ir, const u64* ip, u64 q, u64 a, u64 ap,
                      u64 b, u64 bp) {
    using G = Geom<LOGN, LOGE>;
    static bool attr_set = false;
    if (!attr_set) {
        HX_CHECK(hipFuncSetAttribute((const void*)k_ntt_inv<LOGN, LOGE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_ntt_inv<LOGN, LOGE>), dim3((unsigned)batch), dim3(G::T), G::LDS_BYTES, ctx->stream, x, ir,
                       ip, q, a, ap, b, bp, (u32)batch);
    return (int)hipGetLastError();
}

static int ilog2_exact(u64 n) {
    for (int l = 0; l < 63; ++l)
        if ((1ULL << l) == n) return l;
    return -1;
}

int hx_launch_ntt_fwd(hexl_ctx* ctx, u64* x, size_t batch, const u64* roots, const u64* precon, u64 q, u64 n) {
    if (!batch) return 0;
    const int logn = ilog2_exact(n);
    switch (logn) {
        case 10: return launch_fwd<10, 4>(ctx, x, batch, roots, precon, q);
        case 11: return launch_fwd<11, 5>(ctx, x, batch, roots, precon, q);
        case 12: return launch_fwd<12, 5>(ctx, x, batch, roots, precon, q);
        case 13: return launch_fwd<13, 5>(ctx, x, batch, roots, precon, q);
        case 14:
            return hx_loge_for(14) == 4 ? launch_fwd<14, 4>(ctx, x, batch, roots, precon, q)
                                        : launch_fwd<14, 5>(ctx, x, batch, roots, precon, q);
        default: return HEXL_E_BADARG;
    }
}

int hx_launch_ntt_inv(hexl_ctx* ctx, u64* x, size_t batch, const u64* ir, const u64* ip, u64 q, u64 a, u64 ap,
                      u64 b, u64 bp, u64 n) {
    if (!batch) return 0;
    const int logn = ilog2_exact(n);
    switch (logn) {
        case 10: return launch_inv<10, 4>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 11: return launch_inv<11, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 12: return launch_inv<12, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 13: return launch_inv<13, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        case 14:
            return hx_loge_for(14) == 4 ? launch_inv<14, 4>(ctx, x, batch, ir, ip, q, a, ap, b, bp)
                                        : launch_inv<14, 5>(ctx, x, batch, ir, ip, q, a, ap, b, bp);
        default: return HEXL_E_BADARG;
    }
}
