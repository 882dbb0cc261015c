"""CPU: the keyswitch oracle has no reference vectors to pin it (testdata.zip is an external download),
so it is pinned here by (1) an independent O(n^2) big-integer model of the transforms, (2) a step-by-step
Python-int restatement of SURVEY 2.1-K4, (3) an RLWE correctness check: the output decrypts to t*s_new
under the old key up to small noise -- what the reference's SEAL test asserts end to end
(experimental/bridge-seal/tests/keyswitch-example.cpp:119-206)."""
import numpy as np
import pytest

from ks_util import KsCase


def bitrev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2)


def direct_ntt(a, q, w, n):
    """X[j] = sum_i a_i * w^((2*bitrev(j)+1)*i): HEXL's bit-reversed-output negacyclic NTT"""
    bits = n.bit_length() - 1
    return [sum(int(a[i]) * pow(w, (2 * bitrev(j, bits) + 1) * i, q) for i in range(n)) % q for j in range(n)]


def direct_intt(X, q, w, n):
    bits = n.bit_length() - 1
    winv, ninv = pow(w, -1, q), pow(n, -1, q)
    return [ninv * sum(int(X[j]) * pow(winv, (2 * bitrev(j, bits) + 1) * i, q) for j in range(n)) % q
            for i in range(n)]


@pytest.mark.parametrize("n", [16, 64])
def test_transforms_against_definition(orc, n):
    q = orc.primes(1, 30, n)[0]
    w = orc.orc().orc_minimal_primitive_root(2 * n, q)
    blk = np.zeros(4 * n, dtype=np.uint64)
    orc.orc().orc_tables_keyswitch(n, q, w, orc.p(blk))
    a = orc.splitmix(n, 9, q)
    x = a.copy()
    orc.orc().orc_ks_ntt(orc.p(x), n, q, orc.p(blk[2 * n:3 * n]))
    assert [int(v) for v in x] == direct_ntt(a, q, w, n)
    y = a.copy()
    orc.orc().orc_ks_intt(orc.p(y), n, q, orc.p(blk[0:n]))
    assert [int(v) for v in y] == direct_intt(a, q, w, n)
    # the lazy Harvey transforms (standalone _NTT/_INTT) compute the same function on in-range data
    t = orc.HexlTables(n, q)
    assert np.array_equal(orc.ntt_fwd(a, t)[0], x) and np.array_equal(orc.ntt_inv(a, t)[0], y)


def model_keyswitch(case, t, r, orc):
    """SURVEY 2.1-K4 steps 1-7 with Python ints and the direct transforms above"""
    n, L, K = case.n, case.L, case.K
    qs = [int(v) for v in case.moduli]
    ws = [orc.orc().orc_minimal_primitive_root(2 * n, q) for q in qs]
    sp = K - 1
    c = [direct_intt(t[d * n:(d + 1) * n], qs[d], ws[d], n) for d in range(L)]
    prod = {}
    for i in list(range(L)) + [sp]:
        for k in range(2):
            prod[k, i] = [0] * n
        for d in range(L):
            u = direct_ntt([v % qs[i] for v in c[d]], qs[i], ws[i], n)
            for k in range(2):
                key = case.keys[d][(k * K + i) * n:(k * K + i + 1) * n]
                prod[k, i] = [(p + x * int(y)) % qs[i] for p, x, y in zip(prod[k, i], u, key)]
    out = [int(v) for v in r]
    half = qs[sp] >> 1
    for k in range(2):
        s = [(v + half) % qs[sp] for v in direct_intt(prod[k, sp], qs[sp], ws[sp], n)]
        for i in range(L):
            fix = qs[i] - half % qs[i]
            w_ = direct_ntt([(v + fix) % qs[i] for v in s], qs[i], ws[i], n)
            for j in range(n):
                o = (prod[k, i][j] - w_[j]) * int(case.modswitch[i]) % qs[i]
                out[(k * L + i) * n + j] = (out[(k * L + i) * n + j] + o) % qs[i]
    return np.array(out, dtype=np.uint64)


@pytest.mark.parametrize("n,L,K", [(16, 1, 2), (32, 2, 3), (64, 3, 4), (32, 2, 5)])
def test_oracle_vs_independent_model(orc, n, L, K):
    case = KsCase(orc, n, L, K, seed=n + K, bits=30)
    t, r = case.inputs(orc, 0)
    assert np.array_equal(case.expected(orc, t, r), model_keyswitch(case, t, r, orc))
    case_tw = KsCase(orc, n, L, K, seed=n + K, bits=30, with_twiddles=True)
    assert np.array_equal(case_tw.expected(orc, t, r), case.expected(orc, t, r))


def test_rlwe_decrypts(orc):
    """Build real key-switch keys (special prime P = moduli[K-1]) and check
    result0 + result1*s_old == t*s_new + small noise in every RNS limb."""
    n, L, K = 256, 3, 4
    qs = [int(v) for v in orc.primes(K, 40, n)]
    P = qs[K - 1]
    rng = np.random.default_rng(2)
    s_old = rng.integers(-1, 2, n)
    s_new = rng.integers(-1, 2, n)
    blks = []
    for q in qs:
        b = np.zeros(4 * n, dtype=np.uint64)
        orc.orc().orc_tables_keyswitch(n, q, orc.orc().orc_minimal_primitive_root(2 * n, q), orc.p(b))
        blks.append(b)

    def ntt(poly, i):
        x = np.array([int(v) % qs[i] for v in poly], dtype=np.uint64)
        orc.orc().orc_ks_ntt(orc.p(x), n, qs[i], orc.p(blks[i][2 * n:3 * n]))
        return x.astype(object)

    def intt(x, i):
        y = np.array([int(v) % qs[i] for v in x], dtype=np.uint64)
        orc.orc().orc_ks_intt(orc.p(y), n, qs[i], orc.p(blks[i][0:n]))
        return y.astype(object)

    # key d: limb i holds (b, a) with b = -a*s_old + e + (i == d ? P : 0) * s_new   (NTT domain)
    keys = []
    for d in range(L):
        a_coef = [rng.integers(0, 2**62, n).astype(object) for _ in range(K)]
        e = rng.integers(-3, 4, n)
        key = np.zeros(2 * K * n, dtype=np.uint64)
        for i in range(K):
            a = ntt(a_coef[i] % qs[i], i)
            b = (-a * ntt(s_old, i) + ntt(e, i) + (P % qs[i] if i == d else 0) * ntt(s_new, i)) % qs[i]
            key[(0 * K + i) * n:(0 * K + i + 1) * n] = np.array(b, dtype=np.uint64)
            key[(1 * K + i) * n:(1 * K + i + 1) * n] = np.array(a, dtype=np.uint64)
        keys.append(key)
    moduli = np.array(qs, dtype=np.uint64)
    msf = np.array([pow(P, -1, q) if q != P else 1 for q in qs], dtype=np.uint64)
    # the polynomial being switched, given in NTT form per limb (a consistent RNS element)
    t_int = rng.integers(0, 2**62, n).astype(object)
    t = np.concatenate([np.array(ntt(t_int % qs[d], d), dtype=np.uint64) for d in range(L)])
    res = np.zeros(2 * L * n, dtype=np.uint64)
    orc.keyswitch(res, t, n, L, K, L + 1, moduli, keys, msf)
    noises = []
    for i in range(L):
        r0 = res[(0 * L + i) * n:(0 * L + i + 1) * n].astype(object)
        r1 = res[(1 * L + i) * n:(1 * L + i + 1) * n].astype(object)
        lhs = (r0 + r1 * ntt(s_old, i) - t[i * n:(i + 1) * n].astype(object) * ntt(s_new, i)) % qs[i]
        diff = intt(lhs, i)
        centered = np.array([int(v) if v <= qs[i] // 2 else int(v) - qs[i] for v in diff], dtype=object)
        noises.append(centered)
        assert max(abs(int(v)) for v in centered) < 1 << 24, "keyswitch noise too large in limb %d" % i
    for i in range(1, L):
        assert (noises[i] == noises[0]).all(), "limbs disagree on the noise polynomial"
