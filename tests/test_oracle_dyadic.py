"""CPU: dyadic-multiply oracle vs the reference test's inline expectation and its MultMod."""
import numpy as np

from test_gpu_dyadic import ref_style_io


def test_toy_moduli_exact(orc):
    n, nm, num = 1024, 3, 4
    mod, a, b = ref_style_io(num, nm, n)
    for k in range(num):
        sl = slice(k * 2 * nm * n, (k + 1) * 2 * nm * n)
        m = mod[k * nm:(k + 1) * nm]
        got = orc.dyadic(a[sl], b[sl], n, m).reshape(3, nm, n).astype(object)
        A, B = a[sl].reshape(2, nm, n).astype(object), b[sl].reshape(2, nm, n).astype(object)
        M = m.astype(object)[:, None]
        assert np.array_equal(got[0], A[0] * B[0] % M)
        assert np.array_equal(got[1], (A[0] * B[1] + A[1] * B[0]) % M)
        assert np.array_equal(got[2], A[1] * B[1] % M)
        # the reference's own MultMod agrees on these (tests/test_dyadic_multiply.cpp passes on hardware)
        assert np.array_equal(orc.dyadic(a[sl], b[sl], n, m, exact=False), orc.dyadic(a[sl], b[sl], n, m))


def test_multmod_transliteration_in_domain(orc):
    rng = np.random.default_rng(0)
    n, nm = 4096, 4
    for bits in (30, 45, 52, 61):
        m = np.array(orc.primes(nm, bits, n), dtype=np.uint64)
        a = np.concatenate([rng.integers(0, 4 * int(q), n, dtype=np.uint64) for _ in range(2) for q in m])
        b = np.concatenate([rng.integers(0, 4 * int(q), n, dtype=np.uint64) for _ in range(2) for q in m])
        assert np.array_equal(orc.dyadic(a, b, n, m, exact=False), orc.dyadic(a, b, n, m, exact=True))


def test_the_reference_multmod_is_one_subtraction_short_at_62_bit_moduli(orc):
    """Round 6 (dyadic.hip mulmod_barrett): device/mod_ops.hpp:49-83 estimates the quotient from the product shifted by k - 2
    bits and subtracts q at most ONCE. For q >= 2^61 the estimate can be two short: with q = 4475467519117804091 the product
    4474471586379460888 * 4015622632314414536 comes back as 4564121832859313914 = (x y mod q) + q from the oracle's restatement of that
    function (exact = False), where the mathematical result -- the reference's own test model, tests/test_dyadic_multiply.cpp:59-82, and
    what the GPU kernel returns since round 6 -- is 88654313741509823. Below 2^61 (HEXL's documented range) the two agree on the
    largest operands of every modulus size."""
    import numpy as np
    n = 8
    q = 4475467519117804091
    a = np.zeros(2 * n, dtype=np.uint64)
    b = np.zeros(2 * n, dtype=np.uint64)
    a[0], b[0] = 4474471586379460888, 4015622632314414536
    mod = np.array([q], dtype=np.uint64)
    exact, ref = orc.dyadic(a, b, n, mod, exact=True), orc.dyadic(a, b, n, mod, exact=False)
    assert int(exact[0]) == (int(a[0]) * int(b[0])) % q == 88654313741509823
    assert int(ref[0]) == 4564121832859313914 == int(exact[0]) + q
    rng = np.random.default_rng(3)
    for bits in range(3, 62):
        for q in {(1 << bits) - 1, (1 << (bits - 1)) + 1, int(rng.integers(1 << (bits - 1), 1 << bits))}:
            mod = np.array([q], dtype=np.uint64)
            top = np.array([q - 1 - int(v) % min(q, 5) for v in rng.integers(0, 5, size=2 * n)], dtype=np.uint64)
            mid = rng.integers(0, q, size=2 * n, dtype=np.uint64)
            for x, y in ((top, top), (top, mid), (mid, mid)):
                assert np.array_equal(orc.dyadic(x, y, n, mod, exact=True), orc.dyadic(x, y, n, mod, exact=False)), (bits, q)
