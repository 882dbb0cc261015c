// number_theory.hpp -- host-side number theory for the keyswitch plan (product code, C++).
// Same results as host/src/number_theory_util.cpp (InverseUIntMod :14-46, PowMod :78-92,
// MinimalPrimitiveRoot :134-154) and host/src/twiddle-factors.cpp:16-62 of the reference, which
// feed Device::KeySwitch_load_twiddles / build_invn_meta (host/src/fpga.cpp:1070-1123).
#pragma once
#include <stdint.h>

#include <vector>

namespace hxnt {

typedef uint64_t u64;
typedef unsigned __int128 u128;

inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }

inline u64 powmod(u64 base, u64 e, u64 q) {
    u64 r = 1;
    base %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mulmod(r, base, q);
        base = mulmod(base, base, q);
    }
    return r;
}

inline u64 invmod(u64 a, u64 q) {            // extended Euclid, result in [0,q)
    __int128 t = 0, nt = 1, r = q, nr = a % q;
    while (nr != 0) {
        __int128 k = r / nr, x;
        x = t - k * nt; t = nt; nt = x;
        x = r - k * nr; r = nr; nr = x;
    }
    return (u64)(t < 0 ? t + q : t);
}

// smallest primitive `degree`-th root of unity mod prime q (degree a power of two dividing q-1).
// The reference picks a random primitive root and minimises over its odd powers; the minimum over
// the whole set of primitive roots is unique, so a deterministic search returns the same value.
inline u64 minimal_primitive_root(u64 degree, u64 q) {
    const u64 e = (q - 1) / degree;
    u64 root = 0;
    for (u64 g = 2; g < q && !root; ++g) {
        const u64 c = powmod(g, e, q);
        if (powmod(c, degree / 2, q) == q - 1) root = c;
    }
    const u64 sq = mulmod(root, root, q);
    u64 cur = root, best = root;
    for (u64 i = 0; i < degree / 2; ++i) {   // odd powers root^(2i+1) enumerate all primitive roots
        if (cur < best) best = cur;
        cur = mulmod(cur, sq, q);
    }
    return best;
}

inline u64 bitrev(u64 x, unsigned bits) {
    u64 r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// forward table in bit-reversed order, roots[bitrev(i)] = w^i (twiddle-factors.cpp:25-38)
inline void forward_roots(u64 n, unsigned logn, u64 q, u64 w, u64* roots) {
    roots[0] = 1;
    u64 prev = 0;
    for (u64 i = 1; i < n; ++i) {
        const u64 idx = bitrev(i, logn);
        roots[idx] = mulmod(roots[prev], w, q);
        prev = idx;
    }
}

// hexl-fpga keyswitch inverse layout: stage-major from index 0 (twiddle-factors.cpp:44-55)
inline void inverse_roots_from0(u64 n, u64 q, const u64* roots, u64* inv0) {
    u64 pos = 0;
    for (u64 m = n >> 1; m > 0; m >>= 1)
        for (u64 i = 0; i < m; ++i) inv0[pos++] = invmod(roots[m + i], q);
    inv0[n - 1] = 0;
}

}  // namespace hxnt
