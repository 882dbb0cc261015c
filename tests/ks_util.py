"""Synthetic keyswitch workloads: moduli = GeneratePrimes(K, 51, n) (< 2^52 as
host/src/keyswitch.cpp:32 requires), special prime = moduli[K-1],
modswitch_factors[i] = q_sp^-1 mod q_i, everything else uniform mod its limb (splitmix seeds)."""
import numpy as np


class KsCase:
    def __init__(self, orc, n, L, K, seed=1, bits=51, with_twiddles=False, moduli=None):
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
                    k[(kk * K + i) * n:(kk * K + i + 1) * n] = orc.splitmix(n, seed * 7919 + d * 131 + kk * 17 + i,
                                                                           int(self.moduli[i]))
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

    def expected(self, orc, t, r):
        out = r.copy()
        orc.keyswitch(out, t, self.n, self.L, self.K, self.rns, self.moduli, self.keys, self.modswitch, self.twiddles)
        return out


def primes_below(orc, count, limit, n):
    """the `count` largest primes p < limit with p = 1 (mod 2n), descending"""
    out, v = [], (limit - 2) // (2 * n) * (2 * n) + 1
    while len(out) < count:
        if v < limit and orc.orc().orc_is_prime(v):
            out.append(v)
        v -= 2 * n
    return out
