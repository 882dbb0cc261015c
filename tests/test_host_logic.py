"""CPU: worksize / fence / completion logic of the host API mirror (hexl-fpga_amd/host_api.py), with the
C-ABI replaced by a recording fake whose compute is the oracle. Mirrors the reference's contract in
host/src/fpga_int.cpp:171-507."""
import ctypes

import numpy as np
import pytest

from ks_util import KsCase


class FakeLib:
    """stands in for libhexl_mi355x.so: records every launch, computes with the oracle"""

    def __init__(self, orc):
        self.orc, self.calls = orc, []

    @staticmethod
    def _arr(ptr, count):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(count,))

    def hexl_ntt_fwd_host(self, ctx, xs, batch, roots, precon, q, n):
        self.calls.append(("ntt", batch, q))
        r, p = self._arr(roots, n), self._arr(precon, n)
        for b in range(batch):
            x = self._arr(xs[b], n)
            self.orc.orc().orc_ntt_fwd(self.orc.p(x), n, q, self.orc.p(r), self.orc.p(p))
        return 0

    def hexl_dyadic_multiply_host(self, ctx, outs, a_, b_, batch, n, mods, nm):
        self.calls.append(("dyadic", batch, n, nm))
        for k in range(batch):
            out = self._arr(outs[k], 3 * nm * n)
            out[:] = self.orc.dyadic(self._arr(a_[k], 2 * nm * n).copy(), self._arr(b_[k], 2 * nm * n).copy(), n,
                                     self._arr(mods[k], nm).copy())
        return 0


class FakePlan:
    created = []

    def __init__(self, ctx, n, L, K, rns, kcc, moduli, msf, tw):
        FakePlan.created.append((n, L, K))
        self.args = (n, L, K, rns, np.array(moduli), np.array(msf), tw)
        self.launches = []

    def set_keys(self, keys):
        self.keys = keys

    def keyswitch_host(self, results, ts):
        import orc
        n, L, K, rns, moduli, msf, tw = self.args
        self.launches.append(len(results))
        for r, t in zip(results, ts):
            orc.keyswitch(r, t, n, L, K, rns, moduli, self.keys, msf, tw)

    def close(self):
        pass


@pytest.fixture
def api(hx, orc, monkeypatch):
    fake = FakeLib(orc)
    monkeypatch.setattr(hx, "lib", lambda: fake)
    monkeypatch.setattr(hx, "KeySwitchPlan", FakePlan)
    FakePlan.created.clear()
    a = hx.HexlFpga()
    a._ctx = type("Ctx", (), {"h": None, "close": lambda self: None})()
    a.fake = fake
    return a


def test_worksize_batches_into_one_launch(api, orc):
    n = 1024
    q = orc.primes(1, 30, n)[0]
    t = orc.HexlTables(n, q)
    xs = [orc.splitmix(n, s, q) for s in range(5)]
    work = [x.copy() for x in xs]
    api._set_worksize_NTT(5)
    for w in work:
        api._NTT(w, t.roots, t.precon, q, n)
        assert api.fake.calls == []                 # nothing runs before Completed
    assert api._NTTCompleted() is True
    assert api.fake.calls == [("ntt", 5, q)]
    for w, x in zip(work, xs):
        assert np.array_equal(w, orc.ntt_fwd(x, t)[0])
    # worksize resets to 1: the next call is synchronous
    w = xs[0].copy()
    api._NTT(w, t.roots, t.precon, q, n)
    assert api.fake.calls[-1] == ("ntt", 1, q) and np.array_equal(w, orc.ntt_fwd(xs[0], t)[0])


def test_modulus_change_is_a_fence(api, orc):
    n = 1024
    q1, q2 = orc.primes(2, 30, n)
    t1, t2 = orc.HexlTables(n, q1), orc.HexlTables(n, q2)
    a, b, c = (orc.splitmix(n, s, q1) for s in range(3))
    api._set_worksize_NTT(3)
    api._NTT(a, t1.roots, t1.precon, q1, n)
    api._NTT(b, t1.roots, t1.precon, q1, n)
    api._NTT(c, t2.roots, t2.precon, q2, n)      # flushes the two q1 objects first
    assert api.fake.calls == [("ntt", 2, q1)]
    api._NTTCompleted()
    assert api.fake.calls == [("ntt", 2, q1), ("ntt", 1, q2)]


def test_keyswitch_plan_cache_and_fence(api, orc):
    c1, c2 = KsCase(orc, 64, 2, 3, seed=1, bits=30), KsCase(orc, 64, 1, 2, seed=2, bits=30)
    (t1, r1), (t2, r2) = c1.inputs(orc, 0), c2.inputs(orc, 0)
    e1, e2 = c1.expected(orc, t1, r1), c2.expected(orc, t2, r2)
    # n=64 is below the API's minimum: argument check
    with pytest.raises(ValueError):
        api.KeySwitch(r1, t1, 64, 2, 3, 3, 2, c1.moduli, c1.keys, c1.modswitch)
    # use the checker directly through the fake plan path with a legal n
    c1 = KsCase(orc, 1024, 2, 3, seed=1, bits=30)
    c2 = KsCase(orc, 1024, 1, 2, seed=2, bits=30)
    (t1, r1), (t2, r2) = c1.inputs(orc, 0), c2.inputs(orc, 0)
    e1, e2 = c1.expected(orc, t1, r1), c2.expected(orc, t2, r2)
    api.set_worksize_KeySwitch(3)
    ra, rb = r1.copy(), r1.copy()
    api.KeySwitch(ra, t1, 1024, 2, 3, 3, 2, c1.moduli, c1.keys, c1.modswitch)
    api.KeySwitch(rb, t1, 1024, 2, 3, 3, 2, c1.moduli, c1.keys, c1.modswitch)
    api.KeySwitch(r2, t2, 1024, 1, 2, 2, 2, c2.moduli, c2.keys, c2.modswitch)     # parameter change = fence
    assert FakePlan.created == [(1024, 2, 3)]
    assert api.KeySwitchCompleted() is True
    assert FakePlan.created == [(1024, 2, 3), (1024, 1, 2)]
    assert np.array_equal(ra, e1) and np.array_equal(rb, e1) and np.array_equal(r2, e2)
    # same key pointers again: cached plan, result accumulates
    api.KeySwitch(ra, t1, 1024, 2, 3, 3, 2, c1.moduli, c1.keys, c1.modswitch)
    assert len(FakePlan.created) == 2
    assert np.array_equal(ra, c1.expected(orc, t1, e1))


def test_dyadic_batch(api, orc):
    n, nm = 1024, 2
    mod = np.array([10, 20], dtype=np.uint64)
    a = [np.arange(2 * nm * n, dtype=np.uint64) + k for k in range(3)]
    b = [np.arange(2 * nm * n, dtype=np.uint64) * 3 + k for k in range(3)]
    out = [np.zeros(3 * nm * n, dtype=np.uint64) for _ in range(3)]
    api.set_worksize_DyadicMultiply(3)
    for k in range(3):
        api.DyadicMultiply(out[k], a[k], b[k], n, mod, nm)
    api.DyadicMultiplyCompleted()
    assert api.fake.calls == [("dyadic", 3, n, nm)]
    for k in range(3):
        assert np.array_equal(out[k], orc.dyadic(a[k], b[k], n, mod))


def test_host_accumulate_every_simd_level_matches_numpy(hx):
    """the host's `result += output` (hexl-fpga_amd/csrc/host_simd.cpp; FPGAObject_KeySwitch::fill_out_data, fpga.cpp:441-475): the AVX-512 and
    AVX2 bodies this CPU can run against numpy on canonical words of a 27-, a 52- and a 62-bit modulus, with ragged lengths (vector tails),
    boundary values (sum == q - 1, q, 2q - 2) and unaligned starts. No GPU involved: the symbols are plain host code of the C-ABI library."""
    import ctypes
    hx.build()
    lib = ctypes.CDLL(str(hx.LIB_PATH))
    at_level = getattr(lib, "_Z23hx_add_mod_u64_at_levelPmPKmmmi")
    at_level.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int]
    at_level.restype = None
    auto = getattr(lib, "_Z14hx_add_mod_u64PmPKmmm")
    auto.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64]
    auto.restype = None
    level = getattr(lib, "_Z16hx_add_mod_levelv")
    level.restype = ctypes.c_int
    best = level()
    assert 0 <= best <= 2
    rng = np.random.default_rng(3)
    for q in (133857281, 4503599626682369, 4611686018427322369):
        for n in (1, 7, 16, 33, 16384 + 5):
            r = rng.integers(0, q, n + 1, dtype=np.uint64)
            o = rng.integers(0, q, n + 1, dtype=np.uint64)
            r[1:4] = (q - 1, q - 1, 0)[: max(0, min(3, n))] if n >= 3 else r[1:4]
            o[1:4] = (q - 1, 1, q - 1)[: max(0, min(3, n))] if n >= 3 else o[1:4]
            want = ((r[1:].astype(object) + o[1:].astype(object)) % q).astype(np.uint64)
            for lv in range(best + 1):
                got = r.copy()
                at_level(got[1:].ctypes.data, o[1:].ctypes.data, n, q, lv)        # (start 8 bytes off the allocation's alignment)
                assert got[0] == r[0] and np.array_equal(got[1:], want), (q, n, lv)
            got = r.copy()
            auto(got[1:].ctypes.data, o[1:].ctypes.data, n, q)
            assert np.array_equal(got[1:], want)
