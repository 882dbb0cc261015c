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
    from ks_util import RlweCase
    case = RlweCase(orc, 256, 3, 4, 40)
    res = np.zeros(2 * case.L * case.n, dtype=np.uint64)
    orc.keyswitch(res, case.t, case.n, case.L, case.K, case.L + 1, case.moduli, case.keys, case.modswitch)
    case.check(res)


# ---- the same independent model at the BASELINE modulus size (n = 1024, 51-bit primes, (L, K) in
# {(1,2), (2,3), (6,7)}): the O(n^2) transforms as one object-matrix product per transform
class MatrixModel:
    def __init__(self, orc, case):
        self.case, n = case, case.n
        bits = n.bit_length() - 1
        br = np.array([bitrev(j, bits) for j in range(n)], dtype=np.int64)
        i = np.arange(n, dtype=np.int64)
        self.E = ((2 * br[:, None] + 1) * i[None, :]) % (2 * n)            # exponent of w in row j, column i
        self.qs = [int(v) for v in case.moduli]
        self.fw, self.iv = [], []
        for q in self.qs:
            w = orc.orc().orc_minimal_primitive_root(2 * n, q)
            pw = np.array([pow(w, e, q) for e in range(2 * n)], dtype=object)
            ipw = np.array([pow(w, -e, q) for e in range(2 * n)], dtype=object)
            self.fw.append(pw[self.E])                                     # X = F a
            self.iv.append((ipw[self.E].T, pow(n, -1, q)))                 # a = n^-1 F^-1 X

    def ntt(self, a, i):
        return self.fw[i].dot(np.array([int(v) for v in a], dtype=object)) % self.qs[i]

    def intt(self, X, i):
        M, ninv = self.iv[i]
        return (M.dot(np.array([int(v) for v in X], dtype=object)) * ninv) % self.qs[i]

    def keyswitch(self, t, r):
        case, qs = self.case, self.qs
        n, L, K, sp = case.n, case.L, case.K, case.K - 1
        c = [self.intt(t[d * n:(d + 1) * n], d) for d in range(L)]
        prod = {}
        for i in list(range(L)) + [sp]:
            for k in range(2):
                prod[k, i] = np.zeros(n, dtype=object)
            for d in range(L):
                u = self.ntt(c[d] % qs[i], i)
                for k in range(2):
                    key = case.keys[d][(k * K + i) * n:(k * K + i + 1) * n].astype(object)
                    prod[k, i] = (prod[k, i] + u * key) % qs[i]
        out = r.astype(object)
        half = qs[sp] >> 1
        for k in range(2):
            s = (self.intt(prod[k, sp], sp) + half) % qs[sp]
            for i in range(L):
                fix = qs[i] - half % qs[i]
                w_ = self.ntt((s + fix) % qs[i], i)
                o = (prod[k, i] - w_) * int(case.modswitch[i]) % qs[i]
                sl = slice((k * L + i) * n, (k * L + i + 1) * n)
                out[sl] = (out[sl] + o) % qs[i]
        return np.array([int(v) for v in out], dtype=np.uint64)


@pytest.mark.parametrize("bits", [30, 51, 52])
def test_oracle_vs_independent_models_on_extreme_residues(orc, bits):
    """the worst-case family of the round-6 GPU tests (ks_util.extreme_words: every word at q - 1, beside q / 2, 0 or 1): the oracle
    against the pure-Python model at n = 64 and against the matrix model at n = 1024"""
    from ks_util import primes_below
    for n, L, K, model in ((64, 3, 4, None), (1024, 2, 3, "matrix")):
        moduli = primes_below(orc, K, 1 << bits, n)
        case = KsCase(orc, n, L, K, seed=5, moduli=moduli, extreme_keys=True)
        for b in range(3):
            t, r = case.extreme_inputs(orc, b)
            want = model_keyswitch(case, t, r, orc) if model is None else MatrixModel(orc, case).keyswitch(t, r)
            assert np.array_equal(case.expected(orc, t, r), want)


@pytest.mark.parametrize("L,K", [(1, 2), (2, 3), (6, 7)])
def test_oracle_vs_independent_model_at_baseline_modulus_size(orc, L, K):
    n = 1024
    case = KsCase(orc, n, L, K, seed=51 + L, bits=51)
    t, r = case.inputs(orc, 0)
    assert np.array_equal(case.expected(orc, t, r), MatrixModel(orc, case).keyswitch(t, r))


def test_keyswitch_golden_digests(orc):
    """tests/golden/ks_golden.json (made by tests/golden/make_ks_golden.py from this oracle): the BASELINE shapes'
    outputs must not drift -- the models and RLWE checks above validated exactly these bits"""
    import json
    from pathlib import Path
    gold = json.loads((Path(__file__).parent / "golden" / "ks_golden.json").read_text())
    for v in gold["vectors"]:
        if v["n"] == 16384 and v["instance"] == 1:
            continue                                              # the GPU test covers every vector; keep the CPU suite short
        case = KsCase(orc, v["n"], v["L"], v["K"], seed=v["n"] + v["L"], bits=v["bits"])
        assert [int(x) for x in case.moduli] == v["moduli"]
        t, r = case.inputs(orc, v["instance"])
        assert f"{orc.fnv(t):016x}" == v["fnv_t_target"] and f"{orc.fnv(r):016x}" == v["fnv_result_in"]
        e = case.expected(orc, t, r)
        assert f"{orc.fnv(e):016x}" == v["fnv_result_out"], v
        assert [int(e[0]), int(e[len(e) // 2]), int(e[-1])] == v["out_first_mid_last"]


@pytest.mark.parametrize("L,K", [(6, 7), (7, 8)])
def test_baseline_shape_against_a_coefficient_domain_model(orc, L, K):
    """A third, independent cross-check AT THE BASELINE SHAPE (N = 16384, 51-bit primes; the O(n^2) models above stop at
    n = 1024): the key multiply-accumulate, the mod-up reduction, the special-prime rounding and the mod-down are redone in
    the COEFFICIENT domain with Python integers -- a keyswitch output limb is NTT_i of R(X) = (P_i(X) - t(X)) / q_sp with
    P_i = sum_d (c_d mod q_i) * K_{d,i} (negacyclic product), t = the centred remainder of P_sp modulo q_sp -- on 48 sampled
    coefficients of every output limb. Only the transforms themselves (c_d = INTT(t_d), K = INTT(key), the final INTT of the
    output) are the oracle's, and those are pinned to the reference's own code (tests/test_oracle_golden.py)."""
    n = 16384
    case = KsCase(orc, n, L, K, seed=21)
    qs = [int(v) for v in case.moduli]
    sp, q_sp = K - 1, int(case.moduli[K - 1])
    half = q_sp >> 1
    t, r = case.inputs(orc, 0)
    out = case.expected(orc, t, r)
    blks = []
    for q in qs:
        b = np.zeros(4 * n, dtype=np.uint64)
        orc.orc().orc_tables_keyswitch(n, q, orc.orc().orc_minimal_primitive_root(2 * n, q), orc.p(b))
        blks.append(b)

    def intt(x, i):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        orc.orc().orc_ks_intt(orc.p(y), n, qs[i], orc.p(blks[i][0:n]))
        return y

    c = [intt(t[d * n:(d + 1) * n], d).astype(object) for d in range(L)]                     # step 1, coefficient domain
    rng = np.random.default_rng(3)
    sample = sorted(set([0, 1, n - 1] + [int(v) for v in rng.integers(0, n, 45)]))
    idx = np.arange(n)

    def conv_at(a, b, j, q):
        """coefficient j of a * b mod (X^n + 1, q): sum_{x + y = j} a_x b_y - sum_{x + y = j + n} a_x b_y"""
        y = (j - idx) % n
        sign = np.where(idx <= j, 1, -1).astype(object)
        return int(np.dot(a * sign, b[y])) % q

    # per slot: key polynomials in the coefficient domain, the mod-up of c_d, and the sampled products
    def slot_products(i, k):
        acc = {j: 0 for j in sample}
        for d in range(L):
            kc = intt(case.keys[d][(k * K + i) * n:(k * K + i + 1) * n], i).astype(object)
            cd = c[d] % qs[i]                                                                # intt1_redu.hpp:36-42
            for j in sample:
                acc[j] = (acc[j] + conv_at(cd, kc, j, qs[i])) % qs[i]
        return acc

    for k in range(2):
        S = slot_products(sp, k)
        tc = {j: ((S[j] + half) % q_sp) - half for j in sample}                              # intt2_redu.hpp:25,43,49-51
        for i in range(L):
            P = slot_products(i, k)
            inv = pow(q_sp, -1, qs[i])
            delta = (out[(k * L + i) * n:(k * L + i + 1) * n].astype(object) - r[(k * L + i) * n:(k * L + i + 1) * n].astype(object)) % qs[i]
            got = intt(np.array(delta, dtype=np.uint64), i)
            for j in sample:
                assert int(got[j]) == (P[j] - tc[j]) * inv % qs[i], (k, i, j)                # ms.hpp:70-82


def test_oracle_vs_independent_model_on_random_mixed_moduli(orc):
    """Round 6: the GPU soaks (tools/soak_ks_random.py) compare the kernels with the oracle on random moduli of 18 ... 52 bits in any mix and
    with key_modulus_size up to decomp + 3; here the ORACLE is compared with the pure-Python big-integer model on the same distribution
    (small rings, uniform and worst-case data) -- 1,695 cases of it ran clean in 120 s when this was written, 80 are repeated here"""
    rng = np.random.default_rng(123)

    def random_prime(n, used):
        while True:
            b = int(rng.integers(18, 53))
            v = int(rng.integers(1 << (b - 1), 1 << b)) // (2 * n) * (2 * n) + 1
            while v > (1 << 17) and (v in used or not orc.orc().orc_is_prime(v)):
                v -= 2 * n
            if v > (1 << 17):
                return v

    for _ in range(80):
        n = int(rng.choice([16, 32, 64]))
        L = int(rng.integers(1, 5))
        K = L + 1 + int(rng.integers(0, 3))
        moduli = []
        for _k in range(K):
            moduli.append(random_prime(n, moduli))
        extreme = bool(rng.integers(0, 2))
        case = KsCase(orc, n, L, K, seed=int(rng.integers(1, 1000)), moduli=moduli, extreme_keys=extreme)
        b = int(rng.integers(0, 9))
        t, r = case.extreme_inputs(orc, b) if extreme else case.inputs(orc, b)
        assert np.array_equal(case.expected(orc, t, r), model_keyswitch(case, t, r, orc)), (n, L, K, moduli, extreme)
