"""GPU: the host-side mirror of host/inc/hexl-fpga.h (set_worksize_X / X / XCompleted) end to end with
host pointers -- reads like the reference's gtests (tests/test_fwd_ntt.cpp:27-58,
tests/test_dyadic_multiply.cpp:87-147, tests/test_keyswitch.cpp:122-146)."""
import ctypes

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


# ---- device-resident callers: the same per-object-pointer entry points with device pointers (no staging) ----------

def test_device_pointers_ntt_round_trip(hx, ctx, dev, orc):
    import torch
    n, q = 16384, orc.primes(1, 51, 16384)[0]
    tb = orc.HexlTables(n, q)
    xs = np.stack([stimulus("RANDOM", n, q, seed=40 + s) for s in range(6)])
    d = hx.as_i64(xs).to(dev)                                      # one contiguous device array ...
    lone = hx.as_i64(xs[5].copy()).to(dev)                         # ... and one polynomial somewhere else
    objs = [d[i] for i in range(5)] + [lone]
    lib = hx.lib()
    rc = lib.hexl_ntt_fwd_host(ctx.h, hx.ptr_array(objs), len(objs), tb.roots.ctypes.data, tb.precon.ctypes.data, q, n)
    assert rc == 0
    got = np.vstack([hx.to_u64(d)[:5], hx.to_u64(lone)[None, :]])
    assert np.array_equal(got, orc.ntt_fwd(xs, tb))
    # tables may be device-resident too
    ir, ip = hx.as_i64(tb.inv_roots).to(dev), hx.as_i64(tb.inv_precon).to(dev)
    rc = lib.hexl_ntt_inv_host(ctx.h, hx.ptr_array(objs), len(objs), ir.data_ptr(), ip.data_ptr(), q, tb.inv_n, tb.inv_n_w, n)
    assert rc == 0
    back = np.vstack([hx.to_u64(d)[:5], hx.to_u64(lone)[None, :]])
    assert np.array_equal(back, xs)
    # mixing host and device payload pointers in one call is refused
    host_x = xs[0].copy()
    mixed = (ctypes.c_void_p * 2)(objs[0].data_ptr(), host_x.ctypes.data)
    assert lib.hexl_ntt_fwd_host(ctx.h, mixed, 2, tb.roots.ctypes.data, tb.precon.ctypes.data, q, n) != 0


def test_device_pointers_keyswitch_accumulates_in_place(hx, ctx, dev, orc):
    n, L, K = 4096, 3, 4
    case = KsCase(orc, n, L, K, seed=77)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    ins = [case.inputs(orc, b) for b in range(3)]
    d_t = hx.as_i64(np.concatenate([t for t, _ in ins])).to(dev)
    d_r = hx.as_i64(np.concatenate([r for _, r in ins])).to(dev)
    tt, rs = L * n, 2 * L * n
    # instances 0 and 1 as one contiguous run, then instance 1 AGAIN on the same result array (ordered accumulation),
    # then instance 2
    order = [0, 1, 1, 2]
    plan.keyswitch_host([d_r[b * rs:(b + 1) * rs] for b in order], [d_t[b * tt:(b + 1) * tt] for b in order])
    out = hx.to_u64(d_r).reshape(3, -1)
    e = [case.expected(orc, t, r) for t, r in ins]
    assert np.array_equal(out[0], e[0]) and np.array_equal(out[2], e[2])
    assert np.array_equal(out[1], case.expected(orc, ins[1][0], e[1]))       # applied twice
    plan.close()


def test_device_pointers_dyadic(hx, ctx, dev, orc):
    n, nm, num = 4096, 2, 3
    mod = np.array(orc.primes(nm, 40, n), dtype=np.uint64)
    rng = np.random.default_rng(5)
    def operand():                                                 # [2][n_moduli][n], limb m below moduli[m]
        return np.stack([np.stack([rng.integers(0, int(q), n, dtype=np.uint64) for q in mod]) for _ in range(2)]).reshape(-1)
    a = np.stack([operand() for _ in range(num)])
    b = np.stack([operand() for _ in range(num)])
    d_a, d_b = hx.as_i64(a).to(dev), hx.as_i64(b).to(dev)
    import torch
    d_o = torch.zeros((num, 3 * nm * n), dtype=torch.int64, device=dev)
    lib = hx.lib()
    mods = [mod.copy() for _ in range(num)]
    rc = lib.hexl_dyadic_multiply_host(ctx.h, hx.ptr_array([d_o[i] for i in range(num)]), hx.ptr_array([d_a[i] for i in range(num)]),
                                       hx.ptr_array([d_b[i] for i in range(num)]), num, n, hx.ptr_array(mods), nm)
    assert rc == 0
    got = hx.to_u64(d_o)
    for i in range(num):
        assert np.array_equal(got[i], orc.dyadic(a[i], b[i], n, mod))


@pytest.mark.parametrize("env", [{}, {"HEXL_HOST_ZERO_COPY": "0"}], ids=["zero_copy", "staged"])
def test_lone_keyswitch_through_the_host_entry_point(env):
    """hexl_keyswitch_host at the SEAL bridge's worksize 1 (fpga_context.h:13-16). Round 5: the quarter-transform kernels read t_target from
    and write their output to PINNED HOST memory themselves, publish every quarter limb, and the host adds limb by limb while the rest is in
    flight (capi.hip keyswitch_host_lone); HEXL_HOST_ZERO_COPY=0 keeps the staged route. Same bits as the oracle either way: repeated calls
    (the pinned slabs and completion words are reused), accumulation into the caller's array, two and three objects per call, objects that
    ALIAS one result array (added in order), the 52-bit chain and bridge-seal's mixed one, and L = 7."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch, hexl_fpga_amd as hx, orc
from ks_util import KsCase, seal_chain
ctx = hx.Context(0)
n = 16384
for L, K, moduli in ((6, 7, None), (5, 7, seal_chain(orc, 7, n)), (7, 8, None)):
    case = KsCase(orc, n, L, K, seed=61, moduli=moduli)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch); plan.set_keys(case.keys)
    ins = [case.inputs(orc, b) for b in range(3)]
    for rep in range(4):                                          # one object per call, again and again
        t, r = ins[rep %% 3]
        got = r.copy()
        assert plan.keyswitch_host([got], [t]) is True
        want = case.expected(orc, t, r)
        assert np.array_equal(got, want), ("lone", L, rep)
        assert plan.keyswitch_host([got], [t]) is True            # accumulates into what the first call left
        assert np.array_equal(got, case.expected(orc, t, want)), ("accumulate", L, rep)
    outs = [r.copy() for _, r in ins]
    assert plan.keyswitch_host(outs[:2], [t for t, _ in ins[:2]]) is True
    for b in range(2):
        assert np.array_equal(outs[b], case.expected(orc, *ins[b])), ("two objects", L, b)
    if L < 7:                                                     # three objects at L = 7 exceed the quarter-transform path's default
        shared = ins[0][1].copy()                                 # three objects, ONE result array: added in submission order
        assert plan.keyswitch_host([shared, shared, shared], [t for t, _ in ins]) is True
        want = ins[0][1].copy()
        for t, _ in ins:
            want = case.expected(orc, t, want)
        assert np.array_equal(shared, want), ("aliased", L)
    bad = ins[0][0].copy(); bad[n + 3] = case.moduli[1]           # limb 1, a word equal to its modulus: reported, not fatal
    assert plan.keyswitch_host([ins[0][1].copy()], [bad]) is False
    assert plan.keyswitch_host([ins[0][1].copy()], [ins[0][0]]) is True
    plan.close()
print("OK")
''' % (str(root), str(root / "oracle"), str(root / "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    print(out.stdout[-800:], out.stderr[-2000:])
    assert out.returncode == 0 and out.stdout.strip().endswith("OK")
