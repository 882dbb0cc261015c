"""Synthetic keyswitch workloads: moduli = GeneratePrimes(K, 51, n) (< 2^52 as
host/src/keyswitch.cpp:32 requires), special prime = moduli[K-1],
modswitch_factors[i] = q_sp^-1 mod q_i, everything else uniform mod its limb (splitmix seeds)."""
import numpy as np


class KsCase:
    def __init__(self, orc, n, L, K, seed=1, bits=51, with_twiddles=False, moduli=None, extreme_keys=False):
        self.n, self.L, self.K, self.rns = n, L, K, L + 1
        self.moduli = np.array(orc.primes(K, bits, n) if moduli is None else moduli, dtype=np.uint64)
        q_sp = int(self.moduli[K - 1])
        self.modswitch = np.array([orc.orc().orc_invmod(q_sp % int(q), int(q)) if i < K - 1 else 1
                                   for i, q in enumerate(self.moduli)], dtype=np.uint64)
        self.keys = []
        for d in range(L):
            k = np.empty(2 * K * n, dtype=np.uint64)
            for kk in range(2):
                for i in range(K):
                    k[(kk * K + i) * n:(kk * K + i + 1) * n] = (
                        extreme_words(n, int(self.moduli[i]), d + kk + i) if extreme_keys else
                        orc.splitmix(n, seed * 7919 + d * 131 + kk * 17 + i, int(self.moduli[i])))
            self.keys.append(k)
        self.twiddles = None
        if with_twiddles:
            tw = np.zeros(K * 4 * n, dtype=np.uint64)
            for i in range(K):
                q = int(self.moduli[i])
                orc.orc().orc_tables_keyswitch(n, q, orc.orc().orc_minimal_primitive_root(2 * n, q),
                                               orc.p(tw[i * 4 * n:(i + 1) * 4 * n]))
            self.twiddles = tw
        self.seed = seed

    def inputs(self, orc, b):
        n, L = self.n, self.L
        t = np.concatenate([orc.splitmix(n, self.seed * 104729 + b * 64 + d, int(self.moduli[d])) for d in range(L)])
        r = np.concatenate([orc.splitmix(n, self.seed * 15485863 + b * 64 + k * 32 + i, int(self.moduli[i]))
                            for k in range(2) for i in range(L)])
        return t, r

    def extreme_inputs(self, orc, b):
        """instance b of a worst-case family: every word at an end of the residue range or beside its middle (the largest
        magnitudes of the centred representation the FP64 kernels compute in), in runs and alternations that line signs up"""
        n, L = self.n, self.L
        t = np.concatenate([extreme_words(n, int(self.moduli[d]), b * 5 + d) for d in range(L)])
        r = np.concatenate([extreme_words(n, int(self.moduli[i]), b * 7 + k * 3 + i + 1) for k in range(2) for i in range(L)])
        return t, r

    def expected(self, orc, t, r):
        out = r.copy()
        orc.keyswitch(out, t, self.n, self.L, self.K, self.rns, self.moduli, self.keys, self.modswitch, self.twiddles)
        return out


def extreme_words(n, q, which):
    """n words below q from {q - 1, (q - 1) / 2, (q + 1) / 2, 0, 1} -- constant, alternating, in halves or in runs of 16"""
    lo, hi, top = (q - 1) // 2, (q + 1) // 2, q - 1
    j = np.arange(n)
    pats = [np.full(n, top), np.full(n, lo), np.full(n, hi), np.where(j & 1, hi, lo), np.where(j < n // 2, lo, hi),
            np.where(j & 1, top, 1), np.where((j >> 4) & 1, hi, top), np.where(j & 2, lo, top), np.where(j & 1, 0, hi)]
    return pats[which % len(pats)].astype(np.uint64)


def primes_below(orc, count, limit, n):
    """the `count` largest primes p < limit with p = 1 (mod 2n), descending"""
    out, v = [], (limit - 2) // (2 * n) * (2 * n) + 1
    while len(out) < count:
        if v < limit and orc.orc().orc_is_prime(v):
            out.append(v)
        v -= 2 * n
    return out


def primes_from(orc, count, start, n):
    """the `count` smallest primes p >= start with p = 1 (mod 2n), ascending"""
    out, v = [], (start + 2 * n - 2) // (2 * n) * (2 * n) + 1
    while len(out) < count:
        if orc.orc().orc_is_prime(v):
            out.append(v)
        v += 2 * n
    return out


def seal_chain(orc, K, n):
    """K key moduli shaped like bridge-seal's run (experimental/bridge-seal/tests/seal_test.sh:20: CoeffModulus::Create(16384,
    {52, 30, 30, 40, 27, 27, 27}), SEAL picks the LARGEST primes = 1 mod 2n below each 2^bits): a 52-bit first limb (strict FP64 tier)
    followed by 30 / 30 / 40 / 27 / 27 / 27-bit ones (reduction period 12), the last one being the special prime"""
    bits = [52, 30, 30, 40, 27, 27, 27]
    bits = (bits + bits[1:] * 3)[:K]
    out = []
    for b in bits:
        k = 1 + sum(1 for v in out if v < (1 << b) and v > (1 << (b - 1)))          # the next-largest prime of that size not used yet
        out.append(primes_below(orc, k, 1 << b, n)[k - 1])
    return out


def tier_ladder(orc, K, n):
    """K distinct key moduli that walk through all four FP64 tiers (strict, period 3, 6, 12): the largest unused primes below
    2^52, 2^51, 2^50, 2^49, 2^44, 2^50, ...; the special prime (last) is a 49-bit one"""
    out = []
    for b in ([52, 51, 50, 49, 44, 50, 51, 52] * 2)[:K - 1] + [49]:
        k = 1
        while primes_below(orc, k, 1 << b, n)[k - 1] in out:
            k += 1
        out.append(primes_below(orc, k, 1 << b, n)[k - 1])
    return out


class RlweCase:
    """Real RLWE switching keys with special prime P = moduli[K-1]: key d, limb i holds (b, a) with
    b = -a*s_old + e + (i == d ? P : 0) * s_new in the NTT domain. check(res) asserts that
    res0 + res1*s_old == t*s_new + small noise in every limb and that the limbs agree on the noise polynomial -- what
    the reference's SEAL test asserts end to end (experimental/bridge-seal/tests/keyswitch-example.cpp:119-206)."""

    def __init__(self, orc, n, L, K, bits, seed=2):
        self.orc, self.n, self.L, self.K = orc, n, L, K
        self.qs = [int(v) for v in orc.primes(K, bits, n)]
        qs, P = self.qs, self.qs[K - 1]
        rng = np.random.default_rng(seed)
        self.s_old = rng.integers(-1, 2, n)
        self.s_new = rng.integers(-1, 2, n)
        self.blks = []
        for q in qs:
            b = np.zeros(4 * n, dtype=np.uint64)
            orc.orc().orc_tables_keyswitch(n, q, orc.orc().orc_minimal_primitive_root(2 * n, q), orc.p(b))
            self.blks.append(b)
        self.keys = []
        for d in range(L):
            e = rng.integers(-3, 4, n)
            key = np.zeros(2 * K * n, dtype=np.uint64)
            for i in range(K):
                a = self.ntt(rng.integers(0, 2**62, n).astype(object) % qs[i], i)
                b = (-a * self.ntt(self.s_old, i) + self.ntt(e, i) + (P % qs[i] if i == d else 0) * self.ntt(self.s_new, i)) % qs[i]
                key[(0 * K + i) * n:(0 * K + i + 1) * n] = np.array(b, dtype=np.uint64)
                key[(1 * K + i) * n:(1 * K + i + 1) * n] = np.array(a, dtype=np.uint64)
            self.keys.append(key)
        self.moduli = np.array(qs, dtype=np.uint64)
        self.modswitch = np.array([pow(P, -1, q) if q != P else 1 for q in qs], dtype=np.uint64)
        # the polynomial being switched, given in NTT form per limb (a consistent RNS element)
        t_int = rng.integers(0, 2**62, n).astype(object)
        self.t = np.concatenate([np.array(self.ntt(t_int % qs[d], d), dtype=np.uint64) for d in range(L)])

    def ntt(self, poly, i):
        n, q = self.n, self.qs[i]
        x = np.array([int(v) % q for v in poly], dtype=np.uint64)
        self.orc.orc().orc_ks_ntt(self.orc.p(x), n, q, self.orc.p(self.blks[i][2 * n:3 * n]))
        return x.astype(object)

    def intt(self, x, i):
        n, q = self.n, self.qs[i]
        y = np.array([int(v) % q for v in x], dtype=np.uint64)
        self.orc.orc().orc_ks_intt(self.orc.p(y), n, q, self.orc.p(self.blks[i][0:n]))
        return y.astype(object)

    def check(self, res, noise_bits=24):
        n, L, qs = self.n, self.L, self.qs
        noises = []
        for i in range(L):
            r0 = res[(0 * L + i) * n:(0 * L + i + 1) * n].astype(object)
            r1 = res[(1 * L + i) * n:(1 * L + i + 1) * n].astype(object)
            lhs = (r0 + r1 * self.ntt(self.s_old, i) - self.t[i * n:(i + 1) * n].astype(object) * self.ntt(self.s_new, i)) % qs[i]
            diff = self.intt(lhs, i)
            centered = np.array([int(v) if v <= qs[i] // 2 else int(v) - qs[i] for v in diff], dtype=object)
            noises.append(centered)
            assert max(abs(int(v)) for v in centered) < 1 << noise_bits, "keyswitch noise too large in limb %d" % i
        for i in range(1, L):
            assert (noises[i] == noises[0]).all(), "limbs disagree on the noise polynomial"
