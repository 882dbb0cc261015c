"""GPU parity: hexl_keyswitch (3 fused HIP kernels) vs the CPU oracle, bit-exact, for every n the API
accepts and several (L, K); plus properties at the BASELINE batch size. Mirrors the shape of
tests/test_keyswitch.cpp:148-191 (vectors 16384_6_7_7_2 / 8192_.., one worksize batch)."""
import numpy as np
import pytest

from ks_util import KsCase, primes_below, seal_chain, tier_ladder

pytestmark = pytest.mark.gpu
from pathlib import Path  # noqa: E402
ROOT = Path(__file__).resolve().parent.parent


def run_gpu(hx, ctx, dev, case, ts, rs):
    plan = hx.KeySwitchPlan(ctx, case.n, case.L, case.K, case.rns, 2, case.moduli, case.modswitch, case.twiddles)
    plan.set_keys(case.keys)
    t = hx.as_i64(np.concatenate(ts)).to(dev)
    r = hx.as_i64(np.concatenate(rs)).to(dev)
    plan.keyswitch(r, t, len(ts))
    ctx.sync()
    out = hx.to_u64(r).reshape(len(ts), -1)
    plan.close()
    return out


@pytest.mark.parametrize("n,L,K", [(1024, 1, 2), (1024, 3, 4), (2048, 2, 3), (4096, 5, 7), (8192, 6, 7),
                                   (16384, 6, 7), (16384, 7, 8), (16384, 2, 7), (16384, 15, 16), (32768, 3, 4),
                                   (32768, 6, 7)])
def test_vs_oracle(hx, ctx, dev, orc, n, L, K):
    case = KsCase(orc, n, L, K, seed=n + L)
    nb = 3 if n >= 8192 else 5
    ts, rs = zip(*[case.inputs(orc, b) for b in range(nb)])
    got = run_gpu(hx, ctx, dev, case, ts, rs)
    for b in range(nb):
        assert np.array_equal(got[b], case.expected(orc, ts[b], rs[b])), f"instance {b}"


@pytest.mark.parametrize("which", ["f64_lazy_51bit", "f64_strict_just_below_2^52", "int_55bit", "int_59bit", "int_just_below_2^60",
                                   "int_forced_51bit",
                                   "f64_strict_forced_51bit", "mixed_30_to_52bit", "f64_period6_just_below_2^50",
                                   "f64_period12_just_below_2^49", "f64_period3_forced_on_48bit"])
def test_every_arithmetic_path(hx, ctx, dev, orc, monkeypatch, which):
    """the three kernel families must agree with the oracle: FP64 lazy (all moduli <= 2^51(1+2^-7)), FP64 strict
    (any modulus < 2^52) and the 64-bit integer kernels (moduli up to 2^60, also reachable with HEXL_KS_INT=1)"""
    n, L, K = 16384, 3, 4
    moduli = None
    if which == "f64_strict_just_below_2^52":
        moduli = primes_below(orc, K, 1 << 52, n)
    elif which == "int_55bit":
        moduli = orc.primes(K, 55, n)
    elif which == "int_59bit":                              # the top of the integer kernels' range: 2^59 < q < 2^60
        moduli = orc.primes(K, 59, n)
    elif which == "int_just_below_2^60":
        moduli = primes_below(orc, K, 1 << 60, n)
    elif which == "mixed_30_to_52bit":                      # like the SEAL bridge run: 52,30,30,40,... bit primes
        moduli = [primes_below(orc, 1, 1 << 52, n)[0], orc.primes(1, 30, n)[0], orc.primes(1, 40, n)[0],
                  orc.primes(2, 51, n)[1]]
    elif which == "f64_period6_just_below_2^50":            # fewer range reductions for smaller moduli (f64_arith.hpp)
        moduli = primes_below(orc, K, 1 << 50, n)
    elif which == "f64_period12_just_below_2^49":
        moduli = primes_below(orc, K, 1 << 49, n)
    elif which == "f64_period3_forced_on_48bit":
        moduli = orc.primes(K, 48, n)
        monkeypatch.setenv("HEXL_KS_PERIOD", "3")
    if which == "int_forced_51bit":
        monkeypatch.setenv("HEXL_KS_INT", "1")
    if which == "f64_strict_forced_51bit":
        monkeypatch.setenv("HEXL_KS_NOLAZY", "1")
    case = KsCase(orc, n, L, K, seed=17, moduli=moduli)
    ts, rs = zip(*[case.inputs(orc, b) for b in range(2)])
    got = run_gpu(hx, ctx, dev, case, ts, rs)
    for b in range(2):
        assert np.array_equal(got[b], case.expected(orc, ts[b], rs[b])), f"{which} instance {b}"


@pytest.mark.parametrize("n,L,K,bits", [(1024, 1, 2, 55), (1024, 3, 4, 59), (2048, 2, 3, 57), (4096, 5, 7, 55), (8192, 6, 7, 59),
                                        (16384, 6, 7, 55), (16384, 7, 8, 59), (16384, 15, 16, 53)])
def test_integer_kernels_every_size(hx, ctx, dev, orc, n, L, K, bits):
    """moduli >= 2^52 select the 64-bit integer kernels (keyswitch.hip, second generation: k_ks_intt / k_ksi_special /
    k_ksi_main) at every ring dimension the API accepts for them"""
    case = KsCase(orc, n, L, K, seed=n + L + bits, bits=bits)
    nb = 3 if n >= 8192 else 5
    ts, rs = zip(*[case.inputs(orc, b) for b in range(nb)])
    got = run_gpu(hx, ctx, dev, case, ts, rs)
    for b in range(nb):
        assert np.array_equal(got[b], case.expected(orc, ts[b], rs[b])), f"instance {b}"


@pytest.mark.parametrize("n,L,K", [(1024, 3, 4), (2048, 6, 7), (4096, 2, 3), (8192, 3, 4)])
def test_large_batches_of_small_rings(hx, ctx, dev, orc, n, L, K):
    """batches large enough for the slot-major pipeline at N < 16384 (one workgroup per (instance, limb) has to fill the
    chip twice: keyswitch_x.hip hx_ks_x_applies) and for more than one scratch chunk (256 * 16384 / N instances)"""
    case = KsCase(orc, n, L, K, seed=3 * n + L)
    nb = 2 * 256 * 16384 // (n * L) + 256 * 16384 // n + 37
    ins = [case.inputs(orc, b) for b in range(4)]
    d_t = hx.as_i64(np.concatenate([ins[b % 4][0] for b in range(nb)])).to(dev)
    d_r = hx.as_i64(np.concatenate([ins[b % 4][1] for b in range(nb)])).to(dev)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    plan.keyswitch(d_r, d_t, nb)
    ctx.sync()
    out = hx.to_u64(d_r).reshape(nb, -1)
    plan.close()
    for j in range(4):
        want = case.expected(orc, *ins[j])
        assert (out[j::4] == want).all(), f"instances {j} mod 4"


def test_caller_twiddles_honoured(hx, ctx, dev, orc):
    """twiddle_factors != nullptr path (tests/test_keyswitch.cpp:73-90 passes the 4-block table)"""
    case = KsCase(orc, 4096, 3, 4, seed=11, with_twiddles=True)
    t, r = case.inputs(orc, 0)
    got = run_gpu(hx, ctx, dev, case, [t], [r])
    assert np.array_equal(got[0], case.expected(orc, t, r))


def test_batch_chunks_and_accumulate(hx, ctx, dev, orc):
    """batch larger than one scratch chunk; the output is accumulated into result"""
    n, L, K = 16384, 6, 7
    case = KsCase(orc, n, L, K, seed=5)
    plan = hx.KeySwitchPlan(ctx, n, L, K, L + 1, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    nb = 300                                    # > default chunk of 256
    t0, r0 = case.inputs(orc, 0)
    t1, _ = case.inputs(orc, 1)
    ts = np.concatenate([t0 if b % 2 == 0 else t1 for b in range(nb)])
    rs = np.zeros(nb * 2 * L * n, dtype=np.uint64)
    d_t, d_r = hx.as_i64(ts).to(dev), hx.as_i64(rs).to(dev)
    plan.keyswitch(d_r, d_t, nb)
    ctx.sync()
    out = hx.to_u64(d_r).reshape(nb, -1)
    e0 = case.expected(orc, t0, np.zeros_like(r0))
    e1 = case.expected(orc, t1, np.zeros_like(r0))
    assert np.array_equal(out[0], e0) and np.array_equal(out[1], e1)
    assert np.array_equal(out[298], e0) and np.array_equal(out[299], e1)
    assert (out[0::2] == e0).all() and (out[1::2] == e1).all()
    # additivity in `result` (fpga.cpp:441-475 accumulates): KS(t, r) == r + KS(t, 0) mod q_i, and calling
    # twice adds twice. (KS is NOT linear in t: the special-prime division rounds, intt2_redu.hpp:25-51.)
    qs = np.tile(np.repeat(case.moduli[:L].astype(object), n), 2)
    d_t2, d_r2 = hx.as_i64(t0).to(dev), hx.as_i64(r0).to(dev)
    plan.keyswitch(d_r2, d_t2, 1)
    ctx.sync()
    assert np.array_equal(hx.to_u64(d_r2).astype(object), (r0.astype(object) + e0.astype(object)) % qs)
    plan.keyswitch(d_r2, d_t2, 1)
    ctx.sync()
    assert np.array_equal(hx.to_u64(d_r2).astype(object), (r0.astype(object) + 2 * e0.astype(object)) % qs)
    plan.close()


@pytest.mark.parametrize("n,L,K,nb,strict", [(16384, 7, 8, 80, False), (16384, 7, 8, 80, True), (4096, 3, 4, 176, False)])
def test_fused_and_per_transform_modup_agree(hx, ctx, dev, orc, monkeypatch, n, L, K, nb, strict):
    """large batches run steps 1-2 as one workgroup per input polynomial (k_ksf_up: c stays in registers, L forward
    transforms back to back), small ones as one workgroup per transform; both must give the oracle's bits"""
    if strict:
        monkeypatch.setenv("HEXL_KS_NOLAZY", "1")
    case = KsCase(orc, n, L, K, seed=3 * n + L)
    distinct = [case.inputs(orc, b) for b in range(3)]
    ts = np.concatenate([distinct[b % 3][0] for b in range(nb)])
    rs = np.concatenate([distinct[b % 3][1] for b in range(nb)])
    want = [case.expected(orc, t, r) for t, r in distinct]
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("HEXL_KS_ONE_LANE", "1")             # one chunk of nb: nb * L >= 2 * CUs selects the fused kernel
        plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
        plan.set_keys(case.keys)
        d_t, d_r = hx.as_i64(ts).to(dev), hx.as_i64(rs).to(dev)
        if fuse == "0":
            # the per-transform kernels are what a batch below the threshold gets: run the same data in slices of 8
            for b0 in range(0, nb, 8):
                w = min(8, nb - b0)
                plan.keyswitch(d_r[b0 * 2 * L * n:], d_t[b0 * L * n:], w)
        else:
            plan.keyswitch(d_r, d_t, nb)
        ctx.sync()
        outs.append(hx.to_u64(d_r).reshape(nb, -1))
        plan.close()
    for b in range(nb):
        assert np.array_equal(outs[0][b], want[b % 3]), f"fused, instance {b}"
    assert np.array_equal(outs[0], outs[1])


def test_repeated_launches_are_bit_identical(hx, ctx, dev, orc):
    """the transforms synchronise with one s_barrier and wave-private LDS re-deals: a race would show up as a rare
    run-to-run difference (tools/soak.py runs the long version)"""
    import torch
    n, L, K, nb = 16384, 7, 8, 160
    case = KsCase(orc, n, L, K, seed=21)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    distinct = [case.inputs(orc, b) for b in range(2)]
    ts = np.concatenate([distinct[b % 2][0] for b in range(nb)])
    rs = np.concatenate([distinct[b % 2][1] for b in range(nb)])
    d_t, d_r0 = hx.as_i64(ts).to(dev), hx.as_i64(rs).to(dev)
    ref = None
    for _ in range(12):
        d_r = d_r0.clone()
        plan.keyswitch(d_r, d_t, nb)
        ctx.sync()
        if ref is None:
            ref = d_r.clone()
            out = hx.to_u64(d_r).reshape(nb, -1)
            assert np.array_equal(out[0], case.expected(orc, *distinct[0]))
            assert np.array_equal(out[nb - 1], case.expected(orc, *distinct[(nb - 1) % 2]))
        else:
            assert torch.equal(ref, d_r)
    plan.close()


def test_golden_digests(hx, ctx, dev, orc):
    """the committed digests of the BASELINE shapes (tests/golden/ks_golden.json) straight from the GPU, through both
    pipelines: one launch of the two instances (small-batch kernels) and the two instances inside a batch of 160
    (slot-major pipeline for N = 16384)"""
    import json
    from pathlib import Path
    gold = json.loads((Path(__file__).parent / "golden" / "ks_golden.json").read_text())
    by_shape = {}
    for v in gold["vectors"]:
        by_shape.setdefault((v["n"], v["L"], v["K"], v["bits"]), []).append(v)
    for (n, L, K, bits), vs in by_shape.items():
        case = KsCase(orc, n, L, K, seed=n + L, bits=bits)
        ts, rs = zip(*[case.inputs(orc, v["instance"]) for v in vs])
        for nb in (len(vs), 160):
            tt = [ts[b % len(vs)] for b in range(nb)]
            rr = [rs[b % len(vs)] for b in range(nb)]
            got = run_gpu(hx, ctx, dev, case, tt, rr)
            for b in (0, 1, nb - 2, nb - 1):
                v = vs[b % len(vs)]
                assert f"{orc.fnv(got[b]):016x}" == v["fnv_result_out"], (n, L, K, bits, nb, b)


def test_rlwe_at_baseline_size(hx, ctx, dev, orc):
    """N = 16384, decomp 6 / 7 key moduli, 51-bit primes, REAL switching keys: the GPU output decrypts under the old key to
    t * s_new up to small noise, identically in every limb (what the reference's SEAL bridge test asserts,
    experimental/bridge-seal/tests/keyswitch-example.cpp:119-206) -- and equals the oracle bit for bit"""
    from ks_util import RlweCase
    n, L, K = 16384, 6, 7
    rc = RlweCase(orc, n, L, K, 51)
    zeros = np.zeros(2 * L * n, dtype=np.uint64)
    plan = hx.KeySwitchPlan(ctx, n, L, K, L + 1, 2, rc.moduli, rc.modswitch)
    plan.set_keys(rc.keys)
    for nb in (1, 96):                                            # small-batch kernels, slot-major pipeline
        d_t = hx.as_i64(np.tile(rc.t, nb)).to(dev)
        d_r = hx.as_i64(np.tile(zeros, nb)).to(dev)
        plan.keyswitch(d_r, d_t, nb)
        ctx.sync()
        out = hx.to_u64(d_r).reshape(nb, -1)
        rc.check(out[0])
        assert (out == out[0]).all()
    want = zeros.copy()
    orc.keyswitch(want, rc.t, n, L, K, L + 1, rc.moduli, rc.keys, rc.modswitch)
    assert np.array_equal(out[0], want)
    plan.close()


def test_full_baseline_batch_properties(hx, ctx, dev, orc):
    """BASELINE config 4 at full size: N = 16384, decomp 7, batch 1024 of independent random ciphertexts generated on the
    device. Spot instances (first, both sides of a scratch-chunk boundary, last) against the oracle; every output word in
    range; a second launch on the same inputs adds exactly the same amount (additivity in `result`)."""
    import torch
    n, L, K, nb = 16384, 7, 8, 1024
    case = KsCase(orc, n, L, K, seed=4)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    t = torch.empty((nb, L, n), dtype=torch.int64, device=dev)
    r = torch.empty((nb, 2, L, n), dtype=torch.int64, device=dev)
    for i in range(L):
        t[:, i].random_(0, int(case.moduli[i]), generator=g)
        r[:, :, i].random_(0, int(case.moduli[i]), generator=g)
    r0 = r.clone()
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    plan.keyswitch(r.reshape(-1), t.reshape(-1), nb)
    ctx.sync()
    for b in (0, 255, 256, 1023):
        want = case.expected(orc, hx.to_u64(t[b]).reshape(-1).copy(), hx.to_u64(r0[b]).reshape(-1).copy())
        assert np.array_equal(hx.to_u64(r[b]).reshape(-1), want), f"instance {b}"
    q = torch.tensor([int(v) for v in case.moduli[:L]], dtype=torch.int64, device=dev).view(1, 1, L, 1)
    assert bool(((r >= 0) & (r < q)).all()), "output word out of range"
    delta = (r - r0) % q
    r1 = r.clone()
    plan.keyswitch(r1.reshape(-1), t.reshape(-1), nb)
    ctx.sync()
    assert torch.equal((r1 - r) % q, delta), "second launch added a different amount"
    plan.close()


def test_optional_input_validation(hx, ctx, dev, orc):
    """HEXL_KS_VALIDATE=1 (read once per process, so a child process): a word that is not below its modulus makes the call
    fail with HEXL_E_RANGE and leaves `result` untouched; in-range data passes and matches the oracle"""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch, hexl_fpga_amd as hx, orc
from ks_util import KsCase
dev = torch.device("cuda:0"); ctx = hx.Context(0)
case = KsCase(orc, 4096, 3, 4, seed=9)
plan = hx.KeySwitchPlan(ctx, 4096, 3, 4, 4, 2, case.moduli, case.modswitch); plan.set_keys(case.keys)
t, r = case.inputs(orc, 0)
d_t, d_r = hx.as_i64(t).to(dev), hx.as_i64(r).to(dev)
plan.keyswitch(d_r, d_t, 1); ctx.sync()
assert np.array_equal(hx.to_u64(d_r), case.expected(orc, t, r))
bad = t.copy(); bad[4096 + 17] = case.moduli[1]            # limb 1, exactly q: out of range
d_b, d_r2 = hx.as_i64(bad).to(dev), hx.as_i64(r).to(dev)
try:
    plan.keyswitch(d_r2, d_b, 1)
    print("NOT REJECTED")
except hx.HexlError as e:
    ctx.sync()
    print("REJECTED", "-4" in str(e), bool(np.array_equal(hx.to_u64(d_r2), r)))
''' % (str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HEXL_KS_VALIDATE="1"))
    print(out.stdout[-1000:], out.stderr[-1500:])
    assert out.returncode == 0 and "REJECTED True True" in out.stdout


def test_validation_on_the_host_pointer_path():
    """HEXL_KS_VALIDATE=1 through hexl_keyswitch_host at worksize 1: small sub-batches WRITE their device-side result
    buffer, which is uninitialised on purpose -- the validation must look at t_target only there, accept in-range inputs and compute;
    an out-of-range t_target word is still refused (HEXL_E_RANGE = -4, nothing computed)"""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch, hexl_fpga_amd as hx, orc
from ks_util import KsCase
ctx = hx.Context(0)
case = KsCase(orc, 4096, 3, 4, seed=9)
plan = hx.KeySwitchPlan(ctx, 4096, 3, 4, 4, 2, case.moduli, case.modswitch); plan.set_keys(case.keys)
for rep in range(3):                                            # the staging buffer holds the previous call's output by then
    t, r = case.inputs(orc, rep)
    got = r.copy()
    assert plan.keyswitch_host([got], [t]) is True
    assert np.array_equal(got, case.expected(orc, t, r)), rep
bad = t.copy(); bad[4096 + 17] = case.moduli[1]
r2 = r.copy()
try:
    plan.keyswitch_host([r2], [bad])
    print("NOT REJECTED")
except hx.HexlError as e:
    print("REJECTED", "-4" in str(e), bool(np.array_equal(r2, r)))
''' % (str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HEXL_KS_VALIDATE="1"))
    print(out.stdout[-1000:], out.stderr[-1500:])
    assert out.returncode == 0 and "REJECTED True True" in out.stdout


def test_host_pointer_range_status_covers_its_own_call(hx, ctx, dev, orc):
    """hexl_keyswitch_host returns HEXL_W_RANGE (computed, but an object had a word >= its modulus) for ITS objects only: a flag
    an earlier device-pointer launch left on the plan does not leak into a clean host call, a dirty host call reports
    and still computes the clean objects of the run, and the next clean call is clean again"""
    n, L, K = 4096, 3, 4
    case = KsCase(orc, n, L, K, seed=14)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    t, r = case.inputs(orc, 0)
    bad = t.copy()
    bad[2 * n + 5] = case.moduli[2]
    d_t, d_r = hx.as_i64(bad).to(dev), hx.as_i64(r).to(dev)
    plan.keyswitch(d_r, d_t, 1)                                   # leaves the plan's flag set, nobody reads it
    ctx.sync()
    got = r.copy()
    assert plan.keyswitch_host([got], [t]) is True                # clean call: clean status
    assert np.array_equal(got, case.expected(orc, t, r))
    t1, r1 = case.inputs(orc, 1)
    g0, g1 = r.copy(), r1.copy()
    assert plan.keyswitch_host([g0, g1], [bad, t1]) is False      # dirty object 0, clean object 1
    assert np.array_equal(g1, case.expected(orc, t1, r1))
    got = r.copy()
    assert plan.keyswitch_host([got], [t]) is True
    # the same entry point on DEVICE pointers (the zero-copy branch): a stale flag of an earlier launch is not this call's status either
    # (only the staged branch cleared it), a dirty object of the call itself is
    plan.keyswitch(hx.as_i64(r).to(dev), hx.as_i64(bad).to(dev), 1)
    ctx.sync()
    d_t, d_r = hx.as_i64(t).to(dev), hx.as_i64(r).to(dev)
    assert plan.keyswitch_host([d_r], [d_t]) is True
    assert np.array_equal(hx.to_u64(d_r), case.expected(orc, t, r))
    assert plan.keyswitch_host([hx.as_i64(r).to(dev)], [hx.as_i64(bad).to(dev)]) is False
    assert plan.keyswitch_host([hx.as_i64(r).to(dev)], [d_t]) is True
    plan.close()


@pytest.mark.parametrize("nb", [1, 96, 600])
def test_range_flag_of_the_fp64_kernels(hx, ctx, dev, orc, nb):
    """the FP64 kernels check their precondition (every word below its modulus) where they convert the words: a single
    out-of-range word anywhere in a batch -- in t_target or in result, on the (b, d)-major pipeline (1 instance), the slot-major
    one (96) and across scratch chunks (600) -- shows up in hexl_ks_range_check; clean batches report clean and match the oracle"""
    import torch
    n, L, K = 16384, 3, 4
    case = KsCase(orc, n, L, K, seed=12)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    t, r = case.inputs(orc, 0)
    d_t = hx.as_i64(np.tile(t, nb)).to(dev)
    d_r = hx.as_i64(np.tile(r, nb)).to(dev)
    plan.keyswitch(d_r, d_t, nb)
    assert plan.range_check() is True
    out = hx.to_u64(d_r).reshape(nb, -1)
    want = case.expected(orc, t, r)
    assert np.array_equal(out[0], want) and np.array_equal(out[nb - 1], want)
    for where in ("t", "r"):
        d_t = hx.as_i64(np.tile(t, nb)).to(dev)
        d_r = hx.as_i64(np.tile(r, nb)).to(dev)
        b = nb - 1 if nb > 1 else 0
        if where == "t":                                         # instance b, limb 2, coefficient 77: exactly q
            d_t[b * L * n + 2 * n + 77] = int(case.moduli[2])
        else:                                                     # component 1, limb 0: 2^63 (converts inexactly, still >= q)
            d_r[b * 2 * L * n + (1 * L + 0) * n + 5] = -(1 << 63)
        plan.keyswitch(d_r, d_t, nb)
        assert plan.range_check() is False, where
        assert plan.range_check() is True                         # the check clears the flag
    plan.close()


@pytest.mark.parametrize("env", [{"HEXL_KS_PIPE": "1"}, {"HEXL_KSX_PERSIST": "0"},
                                 {"HEXL_KS_ONE_LANE": "1"}, {"HEXL_KS_INT": "1"}, {"HEXL_KS_INT": "1", "HEXL_KS_PIPE": "1"},
                                 {"HEXL_KS_INT": "1", "HEXL_KSI_LOGE": "4"},
                                 {"HEXL_KS_INT": "1", "HEXL_KSI_LOGE": "4", "HEXL_KS_PIPE": "1"}],
                         ids=["bd_major_pipeline", "slot_major_one_item_per_workgroup", "one_lane",
                              "integer_kernels", "integer_first_generation", "integer_16x1024", "integer_first_generation_16x1024"])
def test_alternative_pipelines_agree_with_the_oracle(env):
    """the kernels the default no longer selects for a large N = 16384 batch -- the (b, d)-major pipeline of round 1
    (k_ksf_up / k_ksf_mac / ...), the slot-major one's non-persistent grids, a single lane,
    the integer kernels in both generations and geometries --
    must still give the oracle's bits (the knobs are read once per process, hence a child process each)"""
    _alternative(env, 16384, 6, 7, 300)


@pytest.mark.parametrize("n,nb", [(1024, 9000), (4096, 2200), (8192, 1100)])
def test_bd_major_pipeline_on_small_rings(n, nb):
    """HEXL_KS_PIPE=1 at N < 16384: the (b, d)-major kernels on batches the slot-major pipeline now takes by default"""
    _alternative({"HEXL_KS_PIPE": "1"}, n, 3, 4, nb)


# Round 4: the slot-major kernels (HEXL_KS_PIPE=3 forces them for every batch) in every arithmetic tier and in both variants of the
# lazy kernels -- SKIP (moduli within a factor 1.25 of each other: c_d and s' enter the transforms without a range reduction, on the
# shifted schedule) and non-SKIP (HEXL_KSX_SKIP=0, or moduli of different sizes) -- plus the strict kernels. Rounds 1-3 covered the
# tiers below 2^51 with two instances only, i.e. on the (b, d)-major kernels.
TIERS = {
    "skip_period3_51bit": ({}, "None"),
    "noskip_forced_period3_51bit": ({"HEXL_KSX_SKIP": "0"}, "None"),
    "skip_period3_ratio_1p24": ({}, "[orc.primes(1, 51, n)[0]] + primes_below(orc, K - 1, int(0.81 * 2**51), n)"),
    "skip_period3_special_prime_smallest": ({}, "orc.primes(K - 1, 51, n) + primes_below(orc, 1, int(0.81 * 2**51), n)"),
    "skip_period6_just_below_2^50": ({}, "primes_below(orc, K, 1 << 50, n)"),
    "skip_period12_just_below_2^49": ({}, "primes_below(orc, K, 1 << 49, n)"),
    "noskip_period6_mixed_50_to_40bit": ({}, "primes_below(orc, 2, 1 << 50, n) + orc.primes(K - 3, 40, n) + orc.primes(1, 45, n)"),
    "noskip_period3_mixed_51_and_30bit": ({}, "orc.primes(2, 51, n)[:1] + orc.primes(K - 2, 30, n) + orc.primes(2, 51, n)[1:]"),
    "strict_just_below_2^52": ({}, "primes_below(orc, K, 1 << 52, n)"),
    # round 6: both sides of the lazy / strict boundary 2^51 (1 + 2^-7) (f64_arith.hpp LAZY_MAX_MODULUS)
    "period3_tier_top_2^51_plus_2^44": ({}, "primes_below(orc, K, (1 << 51) + (1 << 44), n)"),
    "strict_just_above_2^51_plus_2^44": ({}, "primes_from(orc, K, (1 << 51) + (1 << 44), n)"),
    "strict_forced_51bit": ({"HEXL_KS_NOLAZY": "1"}, "None"),
    # round 5: plans whose limbs differ in tier -- every transform takes the tier of ITS modulus (hexl_ks_plan::tier; the reference's NTT
    # engines each run on their own modulus, device/keyswitch/ntt_core.hpp:285-291)
    "mixed_seal_chain_strict_and_period12": ({}, "seal_chain(orc, K, n)"),
    "mixed_seal_chain_plan_wide_tier": ({"HEXL_KS_PER_LIMB": "0"}, "seal_chain(orc, K, n)"),
    "mixed_special_prime_strict_rest_period12": ({}, "orc.primes(K - 1, 47, n) + primes_below(orc, 1, 1 << 52, n)"),
    "mixed_skip_period6_and_period3_around_2^50": ({}, "(primes_below(orc, K, 1 << 50, n) + orc.primes(K, 50, n))[K // 2:K // 2 + K]"),
    "mixed_all_four_tiers": ({}, "tier_ladder(orc, K, n)"),
}
# what plan.tiers() must report for the first K = 4 limbs of the round-5 entries at n = 16384 (test_per_limb_tiers_reported)
MIXED_TIERS_K4 = {
    "mixed_seal_chain_strict_and_period12": ([0, 12, 12, 12], True),
    "mixed_seal_chain_plan_wide_tier": ([0, 0, 0, 0], False),
    "mixed_special_prime_strict_rest_period12": ([12, 12, 12, 0], True),
    "mixed_skip_period6_and_period3_around_2^50": ([6, 6, 3, 3], True),
    "mixed_all_four_tiers": ([0, 3, 6, 12], True),
}


@pytest.mark.parametrize("tier", list(TIERS))
@pytest.mark.parametrize("n,L,K,nb", [(16384, 6, 7, 70), (16384, 3, 4, 300), (2048, 3, 4, 40), (32768, 3, 4, 5)])
def test_slot_major_kernels_in_every_tier(tier, n, L, K, nb):
    env, moduli = TIERS[tier]
    _alternative(dict(env, HEXL_KS_PIPE="3"), n, L, K, nb, moduli)


# Round 6: the X / I reduction schedules leave sums un-reduced wherever the worst-case bound chain allows it. Uniform inputs stay far
# from those bounds; these do not try to reach them (no input does on all 14 stages) but start every chain at its largest magnitudes:
# every word of the keys, t_target and result at q - 1, beside q / 2, 0 or 1, constant / alternating / in runs (ks_util.extreme_words)
EXTREME_TIERS = ["skip_period3_51bit", "period3_tier_top_2^51_plus_2^44", "strict_just_above_2^51_plus_2^44", "skip_period6_just_below_2^50",
                 "skip_period12_just_below_2^49", "strict_just_below_2^52", "mixed_seal_chain_strict_and_period12", "mixed_all_four_tiers"]


@pytest.mark.parametrize("tier", EXTREME_TIERS)
@pytest.mark.parametrize("n,L,K,nb,env", [(16384, 3, 4, 300, {"HEXL_KS_PIPE": "3"}), (16384, 3, 4, 9, {"HEXL_KS_LAT": "0"}),
                                          (16384, 3, 4, 3, {"HEXL_KS_LAT": "2"}), (32768, 3, 4, 90, {})])
def test_extreme_residues_in_every_tier(tier, n, L, K, nb, env):
    """slot-major kernels in every tier; the (b, d)-major kernels, the quarter-transform latency path and the N = 32768 halves in three"""
    if "HEXL_KS_PIPE" not in env and tier not in ("period3_tier_top_2^51_plus_2^44", "strict_just_below_2^52",
                                                  "mixed_seal_chain_strict_and_period12"):
        pytest.skip("pipeline covered in three tiers")
    tenv, moduli = TIERS[tier]
    _alternative(dict(tenv, **env), n, L, K, nb, moduli, extreme=True)


@pytest.mark.parametrize("L,K,nb,env", [(3, 4, 80, {}), (6, 7, 40, {}), (1, 2, 230, {}), (15, 16, 3, {"HEXL_KS_PIPE": "3"}),
                                         (2, 7, 5, {"HEXL_KS_PIPE": "3"}), (3, 4, 80, {"HEXL_KS_PIPE": "1"})])
def test_n32768_on_the_slot_major_pipeline(L, K, nb, env):
    """Round 5: at N = 32768 every transform of the slot-major pipeline is TWO 16384-point halves (keyswitch_x.hip k_ksh_*:
    the outermost stage is a radix-2 step across the halves, done on load by the consumer / by the finishing pass of an inverse; one
    workgroup per (instance, limb, half)). Batches that fill the chip take it by default (nb x L >= 224), HEXL_KS_PIPE=3 forces it,
    HEXL_KS_PIPE=1 keeps the (b, d)-major kernels: same bits as the oracle on all of them, chunk boundaries included (128 per chunk)."""
    _alternative(env, 32768, L, K, nb)


@pytest.mark.parametrize("tier", list(TIERS))
@pytest.mark.parametrize("L,K", [(6, 7), (7, 8), (3, 4), (1, 2), (5, 7), (15, 16)])
def test_lone_keyswitch_latency_path(tier, L, K):
    """keyswitch_lat.hip: a lone keyswitch at N = 16384 runs every transform as four quarter transforms on four compute units
    (HEXL_KS_LAT=2 sends every instance of a batch down that path, one by one): same bits as the oracle in every tier"""
    env, moduli = TIERS[tier]
    if (L, K) in ((5, 7), (15, 16)) and tier not in ("skip_period3_51bit", "strict_just_below_2^52", "noskip_forced_period3_51bit",
                                                      "mixed_seal_chain_strict_and_period12"):
        pytest.skip("shape covered in four tiers")
    if K < 3 and "mixed_50_to_40bit" in tier:
        pytest.skip("the mixed tier needs three key moduli")
    _alternative(dict(env, HEXL_KS_LAT="2"), 16384, L, K, 3, moduli)


@pytest.mark.parametrize("tier", [t for t in TIERS if t.startswith("mixed_")])
@pytest.mark.parametrize("n,L,K,nb,env", [(16384, 6, 7, 5, {"HEXL_KS_LAT": "0"}),                       # five kernels, one transform per workgroup
                                          (16384, 3, 4, 160, {"HEXL_KS_PIPE": "1"}),                    # large batch kept off the slot-major pipeline
                                          (16384, 6, 7, 2, {"HEXL_KS_LAT": "1"}),                       # the three-kernel path (k_ksl_*)
                                          (16384, 6, 7, 2, {}),                                          # two instances in one launch of the quarter-transform kernels
                                          (32768, 3, 4, 6, {}), (4096, 3, 4, 9, {"HEXL_KS_LAT": "0"})])
def test_bd_major_kernels_with_limbs_of_different_tiers(tier, n, L, K, nb, env):
    """keyswitch_f64.hip: the (b, d)-major kernels built with LAZY = -1 look the reduction schedule up per transform (with_tier;
    HEXL_KS_PER_LIMB=2 -- by default this pipeline runs plans of mixed tiers on the plan-wide tier, which measured faster at the batch
    sizes it serves: the second run)"""
    tenv, moduli = TIERS[tier]
    if tenv.get("HEXL_KS_PER_LIMB") != "0":
        _alternative(dict(tenv, **env, HEXL_KS_PER_LIMB="2"), n, L, K, nb, moduli)
    _alternative(dict(tenv, **env), n, L, K, nb, moduli)


def test_per_limb_tiers_reported(hx, ctx, orc):
    """hexl_ks_plan_tiers: the tier every limb's transforms run in, and whether the limbs in use differ"""
    n, L, K = 16384, 3, 4
    for name, (want, mixed) in MIXED_TIERS_K4.items():
        env, expr = TIERS[name]
        if env:
            continue                                              # (environment knobs are read per plan, but keep this test in-process)
        moduli = eval(expr, {"orc": orc, "n": n, "K": K, "primes_below": primes_below, "seal_chain": seal_chain, "tier_ladder": tier_ladder})
        case = KsCase(orc, n, L, K, seed=3, moduli=moduli)
        plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
        assert plan.tiers() == (want, mixed), name
        plan.close()
    case = KsCase(orc, n, L, K, seed=3)                           # BASELINE's primes: one tier, nothing mixed
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    assert plan.tiers() == ([3, 3, 3, 3], False)
    plan.close()
    case = KsCase(orc, n, L, K, seed=3, bits=55)                  # integer kernels
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    assert plan.tiers() == ([-1, -1, -1, -1], False)
    plan.close()


def test_latency_paths_agree_on_one_keyswitch(hx, ctx, dev, orc):
    """the default for one instance (quarter transforms), in-process, incl. accumulation into a non-zero result and a second call"""
    n, L, K = 16384, 6, 7
    case = KsCase(orc, n, L, K, seed=33)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    for b in range(3):
        t, r = case.inputs(orc, b)
        d_t, d_r = hx.as_i64(t).to(dev), hx.as_i64(r).to(dev)
        plan.keyswitch(d_r, d_t, 1)
        ctx.sync()
        want = case.expected(orc, t, r)
        assert np.array_equal(hx.to_u64(d_r), want), b
        plan.keyswitch(d_r, d_t, 1)                               # accumulates again
        ctx.sync()
        assert np.array_equal(hx.to_u64(d_r), case.expected(orc, t, want)), b
    plan.close()


@pytest.mark.parametrize("env", [{}, {"HEXL_KSX_PERSIST": "0"}, {"HEXL_KS_PER_LIMB": "0"}], ids=["default", "one_workgroup_per_item", "plan_wide_tier"])
def test_inverse_after_inverse_race_many_small_workgroups(env):
    """Round 6, found by tools/soak_ks_random.py: an inverse transform ends with cross-wave LDS reads, and the wave-private first re-deal
    of an inverse transform that FOLLOWS it in the same workgroup (k_ksx_special's second one) wrote a fast wave's block while a slow wave
    was still reading it -- at N = 2048 (eight two-wave workgroups per CU, per-pass wave priorities) 9-27 of 180,000 instances came back
    with a wrong k = 0 half. ntt_core.hpp ReadersGate closes it; tools/race_probe.py compares every instance of 40 launches of 4,500."""
    import os
    import subprocess
    import sys
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "race_probe.py"), "2048", "2", "3", "4500", "40"], capture_output=True,
                         text=True, timeout=600, env=dict(os.environ, **env))
    print(out.stdout[-600:], out.stderr[-800:])
    assert out.returncode == 0 and " 0 wrong instances" in out.stdout


def _alternative(env, n, L, K, nb, moduli="None", extreme=False):
    """`nb` instances (three distinct ones repeated) through the library in a child process with `env` set, against the oracle
    (`extreme`: keys and inputs from ks_util.extreme_words instead of uniform ones)"""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch, hexl_fpga_amd as hx, orc
from ks_util import KsCase, primes_below, primes_from, seal_chain, tier_ladder
dev = torch.device("cuda:0"); ctx = hx.Context(0)
n, L, K, nb = %d, %d, %d, %d
extreme = %s
case = KsCase(orc, n, L, K, seed=77, moduli=%s, extreme_keys=extreme)
plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch); plan.set_keys(case.keys)
ins = [case.extreme_inputs(orc, b) if extreme else case.inputs(orc, b) for b in range(3)]
d_t = hx.as_i64(np.concatenate([ins[b %% 3][0] for b in range(nb)])).to(dev)
d_r = hx.as_i64(np.concatenate([ins[b %% 3][1] for b in range(nb)])).to(dev)
plan.keyswitch(d_r, d_t, nb); ctx.sync()
out = hx.to_u64(d_r).reshape(nb, -1)
want = [case.expected(orc, t, r) for t, r in ins]
print("OK" if all(np.array_equal(out[b], want[b %% 3]) for b in range(nb)) else "MISMATCH")
''' % (str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests"), n, L, K, nb, extreme, moduli)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    print(out.stdout[-500:], out.stderr[-1500:])
    assert out.returncode == 0 and out.stdout.strip().endswith("OK")
