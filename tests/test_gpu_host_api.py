"""GPU: the host-side mirror of host/inc/hexl-fpga.h (set_worksize_X / X / XCompleted) end to end with
host pointers -- reads like the reference's gtests (tests/test_fwd_ntt.cpp:27-58,
tests/test_dyadic_multiply.cpp:87-147, tests/test_keyswitch.cpp:122-146)."""
import numpy as np
import pytest

from conftest import stimulus
from ks_util import KsCase

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(hx):
    a = hx.HexlFpga()
    a.acquire_FPGA_resources()
    yield a
    a.release_FPGA_resources()


def test_ntt_intt_worksize_batches(api, orc):
    n = 16384
    q = orc.primes(1, 55, n)[0]
    t = orc.HexlTables(n, q)
    xs = [stimulus("RANDOM", n, q, seed=s) for s in range(8)]
    work = [x.copy() for x in xs]
    api._set_worksize_NTT(len(work))
    for w in work:
        api._NTT(w, t.roots, t.precon, q, n)
    assert api._NTTCompleted()
    for w, x in zip(work, xs):
        assert np.array_equal(w, orc.ntt_fwd(x, t)[0])
    api._set_worksize_INTT(len(work))
    for w in work:
        api._INTT(w, t.inv_roots, t.inv_precon, q, t.inv_n, t.inv_n_w, n)
    assert api._INTTCompleted()
    for w, x in zip(work, xs):
        assert np.array_equal(w, x)
    # worksize 1: completes inside the call (fpga_int.cpp:459-461)
    w = xs[0].copy()
    api._NTT(w, t.roots, t.precon, q, n)
    assert np.array_equal(w, orc.ntt_fwd(xs[0], t)[0])


def test_ntt_fence_on_modulus_change(api, orc):
    n = 16384
    q1, q2 = orc.primes(2, 32, n)
    t1, t2 = orc.HexlTables(n, q1), orc.HexlTables(n, q2)
    a, b = stimulus("RANDOM", n, q1, 1), stimulus("RANDOM", n, q2, 2)
    wa, wb = a.copy(), b.copy()
    api._set_worksize_NTT(2)
    api._NTT(wa, t1.roots, t1.precon, q1, n)
    api._NTT(wb, t2.roots, t2.precon, q2, n)
    api._NTTCompleted()
    assert np.array_equal(wa, orc.ntt_fwd(a, t1)[0]) and np.array_equal(wb, orc.ntt_fwd(b, t2)[0])


def test_dyadic_multiply_batches(api, orc):
    n, nm, num = 4096, 2, 4
    mod = np.array(orc.primes(nm, 40, n), dtype=np.uint64)
    rng = np.random.default_rng(0)
    ops = [(np.concatenate([rng.integers(0, int(m), n, dtype=np.uint64) for _ in range(2) for m in mod]),
            np.concatenate([rng.integers(0, int(m), n, dtype=np.uint64) for _ in range(2) for m in mod]))
           for _ in range(num)]
    outs = [np.zeros(3 * nm * n, dtype=np.uint64) for _ in range(num)]
    api.set_worksize_DyadicMultiply(num)
    for o, (a, b) in zip(outs, ops):
        api.DyadicMultiply(o, a, b, n, mod, nm)
    assert api.DyadicMultiplyCompleted()
    for o, (a, b) in zip(outs, ops):
        assert np.array_equal(o, orc.dyadic(a, b, n, mod))


def test_keyswitch_batch_and_key_cache(api, orc):
    case = KsCase(orc, 8192, 6, 7, seed=3)
    ins = [case.inputs(orc, b) for b in range(4)]
    res = [r.copy() for _, r in ins]
    api.set_worksize_KeySwitch(len(ins))
    for r, (t, _) in zip(res, ins):
        api.KeySwitch(r, t, case.n, case.L, case.K, case.rns, 2, case.moduli, case.keys, case.modswitch)
    assert api.KeySwitchCompleted()
    for r, (t, r0) in zip(res, ins):
        assert np.array_equal(r, case.expected(orc, t, r0))
    # second batch re-uses the cached plan (same key pointers) and accumulates into result again
    r = res[0]
    before = r.copy()
    api.KeySwitch(r, ins[0][0], case.n, case.L, case.K, case.rns, 2, case.moduli, case.keys, case.modswitch)
    assert np.array_equal(r, case.expected(orc, ins[0][0], before))
    assert len(api._plans) == 1


def test_argument_checks(api, orc):
    x = np.zeros(512, dtype=np.uint64)
    with pytest.raises(ValueError):
        api._NTT(x, x, x, 97, 512)
    with pytest.raises(ValueError):
        api.KeySwitch(x, x, 16384, 6, 7, 7, 3, x, [x], x)
