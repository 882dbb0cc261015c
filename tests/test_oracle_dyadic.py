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
