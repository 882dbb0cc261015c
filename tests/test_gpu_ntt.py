"""GPU parity: hexl_ntt_fwd / hexl_ntt_inv (HIP, through the C-ABI) vs the CPU oracle.
Mirrors tests/test_fwd_ntt.cpp:119-170 and tests/test_inv_ntt.cpp:127-178 of the reference:
N=16384, primes of 20/32/55/62 "bits", stimuli RANDOM/RAMP/ZEROS/ONES/IMPULSE/ALL_MAX, exact
equality (bit-exact, including uint64 wrap-around for out-of-range inputs and 4q >= 2^64)."""
import json
from pathlib import Path

import numpy as np
import pytest

from conftest import stimulus

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / "golden" / "ntt_golden.json").read_text())["records"]
STIMS = ["RANDOM", "RAMP", "ALL_ZEROS", "ALL_ONES", "IMPULSE", "ALL_MAX_VALUES"]


def _dev(hx, a, dev):
    return hx.as_i64(a).to(dev)


def run_fwd(hx, ctx, dev, x, t):
    d = _dev(hx, x, dev)
    ctx.ntt_fwd(d, _dev(hx, t.roots, dev), _dev(hx, t.precon, dev), t.q, t.n)
    ctx.sync()
    return hx.to_u64(d).reshape(-1, t.n)


def run_inv(hx, ctx, dev, x, t):
    d = _dev(hx, x, dev)
    ctx.ntt_inv(d, _dev(hx, t.inv_roots, dev), _dev(hx, t.inv_precon, dev), t.q, t.inv_n, t.inv_n_w, t.n)
    ctx.sync()
    return hx.to_u64(d).reshape(-1, t.n)


@pytest.mark.parametrize("bits", [20, 32, 55, 62])
def test_fwd_inv_reference_matrix(hx, ctx, dev, orc, bits):
    n = 16384
    q = orc.primes(1, bits, n)[0]
    t = orc.HexlTables(n, q)
    x = np.stack([stimulus(k, n, q) for k in STIMS])
    got = run_fwd(hx, ctx, dev, x, t)
    exp = orc.ntt_fwd(x, t)
    for k, name in enumerate(STIMS):
        assert np.array_equal(got[k], exp[k]), f"fwd {bits}-bit {name}"
    got = run_inv(hx, ctx, dev, x, t)
    exp = orc.ntt_inv(x, t)
    for k, name in enumerate(STIMS):
        assert np.array_equal(got[k], exp[k]), f"inv {bits}-bit {name}"


@pytest.mark.parametrize("rec", [r for r in GOLD], ids=lambda r: f"n{r['n']}_b{r['bits']}")
def test_golden_digests(hx, ctx, dev, orc, rec):
    """committed fixtures captured from the reference's own CPU oracle"""
    n, q = rec["n"], rec["q"]
    t = orc.HexlTables(n, q)
    assert t.w == rec["w"] and t.inv_n == rec["inv_n"] and t.inv_n_w == rec["inv_n_w"]
    for name, s in rec["stimuli"].items():
        x = {"RAMP": np.arange(n, dtype=np.uint64), "ALLMAX": np.full(n, 2**64 - 1, dtype=np.uint64),
             "SPLITMIX42": orc.splitmix(n, 42, q)}[name]
        f = run_fwd(hx, ctx, dev, x, t)[0]
        assert "%016x" % orc.fnv(f) == s["fwd_fnv"], f"fwd {name}"
        assert [int(v) for v in f[:4]] == s["fwd_head"] and [int(v) for v in f[-4:]] == s["fwd_tail"]
        i = run_inv(hx, ctx, dev, x, t)[0]
        assert "%016x" % orc.fnv(i) == s["inv_fnv"], f"inv {name}"
        assert [int(v) for v in i[:4]] == s["inv_head"] and [int(v) for v in i[-4:]] == s["inv_tail"]


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384])
def test_all_sizes_random(hx, ctx, dev, orc, n):
    q = orc.primes(2, 51, n)[1]
    t = orc.HexlTables(n, q)
    x = np.stack([orc.splitmix(n, 100 + b, q) for b in range(5)])
    assert np.array_equal(run_fwd(hx, ctx, dev, x, t), orc.ntt_fwd(x, t))
    assert np.array_equal(run_inv(hx, ctx, dev, x, t), orc.ntt_inv(x, t))


@pytest.mark.parametrize("limit", [(1 << 52) + (1 << 49), 1 << 52, (1 << 51) + (1 << 44), 1 << 50, 1 << 49, 1 << 30],
                         ids=["strict_top_2^52x1.125", "strict_2^52", "period3_top", "period6_top", "period12_top", "30bit"])
@pytest.mark.parametrize("n", [16384, 32768, 2048])
def test_extreme_residues_at_every_tier_top(hx, ctx, dev, orc, limit, n):
    """Round 6 (X / I reduction schedules): the largest prime of every FP64 tier, polynomials whose words all sit at q - 1, beside q / 2,
    at 0 or 1 -- constant, alternating, in halves and runs (ks_util.extreme_words) --, in a batch large enough for the persistent kernels"""
    from ks_util import extreme_words, primes_below
    q = primes_below(orc, 1, limit, n)[0]
    t = orc.HexlTables(n, q)
    base = np.stack([extreme_words(n, q, k) for k in range(9)])
    batch = 300 * 16384 // n + 4
    x = base[np.arange(batch) % 9].copy()
    for run, ref in ((run_fwd, orc.ntt_fwd), (run_inv, orc.ntt_inv)):
        got, want = run(hx, ctx, dev, x, t), ref(base, t)
        assert (got == want[np.arange(batch) % 9]).all()


@pytest.mark.parametrize("n,batch", [(2048, 12000), (2048, 5600), (4096, 6000), (8192, 3000), (16384, 1500)])
def test_every_polynomial_of_large_persistent_batches(hx, ctx, dev, orc, n, batch):
    """Round 6 (tools/soak_ntt_random.py): the persistent inverse kernel runs transform after transform in one workgroup, and the first
    (wave-private) re-deal of a transform overwrote words a slower wave was still reading out of the previous one's cross-wave exchange --
    at N = 2048, batches of 5,600 / 12,000, one or two polynomials of a launch came back wrong about once a second. Rounds 1-5 compared
    a sample of each large batch; this compares EVERY polynomial of ten launches (on the device). ntt_core.hpp ReadersGate."""
    import torch
    q = orc.primes(2, 51, n)[1]
    t = orc.HexlTables(n, q)
    base = np.stack([orc.splitmix(n, 500 + b, q) for b in range(7)])
    idx = torch.arange(batch, device=dev) % 7
    x = torch.from_numpy(base.view(np.int64)).to(dev)[idx].contiguous()
    tabs = [_dev(hx, a, dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]
    for fwd in (False, True):
        want = torch.from_numpy((orc.ntt_fwd if fwd else orc.ntt_inv)(base, t).view(np.int64)).to(dev)[idx]
        for launch in range(10):
            d = x.clone()
            if fwd:
                ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
            else:
                ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n)
            ctx.sync()
            wrong = int((d.view(batch, n) != want).any(dim=1).sum())
            assert wrong == 0, f"{'forward' if fwd else 'inverse'} launch {launch}: {wrong} of {batch} polynomials differ"


def test_random_tables_like_benchmark(hx, ctx, dev, orc):
    """benchmark/bench_fwd_ntt.cpp:38-42 feeds RANDOM tables: the kernel must replay the butterflies
    op for op, not rely on the tables being roots of unity"""
    n, q = 16384, orc.primes(1, 52, 16384)[0]
    t = orc.HexlTables(n, q)
    rng = np.random.default_rng(3)
    for arr in (t.roots, t.precon, t.inv_roots, t.inv_precon):
        arr[:] = rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)
    x = rng.integers(0, 2**64 - 1, size=(3, n), dtype=np.uint64)
    assert np.array_equal(run_fwd(hx, ctx, dev, x, t), orc.ntt_fwd(x, t))
    t.inv_n, t.inv_n_w = int(rng.integers(0, q)), int(rng.integers(0, q))
    assert np.array_equal(run_inv(hx, ctx, dev, x, t), orc.ntt_inv(x, t))


def test_reference_benchmark_workload_takes_the_integer_path_once(orc):
    """benchmark/bench_fwd_ntt.cpp:25-42 / bench_inv_ntt.cpp: q = 136314881 (< 2^52, so the FP64 fast-path kernels are launched) with
    RANDOM roots / precons / inv_n / inv_n_w: the tables fail the device-side Shoup check, which is known at kernel entry -- the
    kernels must go straight to the integer butterflies (rounds 2-3 ran the FP64 transform first and then fell back: the transform
    twice). Bit-exact against the oracle, and not slower than the integer-only kernels (HEXL_NTT_INT=1) on the same data."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys, json
sys.path[:0] = [%r, %r]
import numpy as np, torch, hexl_fpga_amd as hx, orc
dev = torch.device("cuda:0"); ctx = hx.Context(0)
n, q, batch = 16384, 136314881, 1024
rng = np.random.default_rng(5)
t = orc.HexlTables(n, q)
for arr in (t.roots, t.precon, t.inv_roots, t.inv_precon):
    arr[:] = rng.integers(0, q, size=n, dtype=np.uint64)
t.inv_n, t.inv_n_w = int(rng.integers(0, q)), int(rng.integers(0, q))
x = rng.integers(0, q, size=(8, n), dtype=np.uint64)
d_in = hx.as_i64(np.tile(x, (batch // 8, 1)).reshape(-1)).to(dev)
d = d_in.clone()
tabs = [hx.as_i64(a).to(dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]
res = {}
for name in ("fwd", "inv"):
    run = (lambda: ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)) if name == "fwd" else (lambda: ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n))
    d.copy_(d_in)
    run(); ctx.sync()                                         # (the first call also leaves the host its hint: ntt.hip NttHint)
    got = hx.to_u64(d).reshape(batch, n)
    want = orc.ntt_fwd(x, t) if name == "fwd" else orc.ntt_inv(x, t)
    res[name + "_exact"] = bool(np.array_equal(got[:8], want) and np.array_equal(got[-8:], want))
    d.copy_(d_in)
    run(); ctx.sync()                                         # second call: the hinted route, same bits
    res[name + "_exact"] = res[name + "_exact"] and bool(np.array_equal(hx.to_u64(d).reshape(batch, n)[-8:], want))
    for _ in range(20): run()
    torch.cuda.synchronize()
    best = None                                               # best of four blocks of 50 launches: a stalled launch thread on a shared
    for _ in range(4):                                        # host drains the queue and would otherwise land in the device time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        best = ms if best is None else min(best, ms)
    res[name + "_ms"] = best
print("RESULT", json.dumps(res))
''' % (str(root), str(root / "oracle"))
    out = {}
    for label, env in (("default", {}), ("integer_only", {"HEXL_NTT_INT": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stderr[-1500:]
        out[label] = json.loads(line[-1][7:])
    print(out)
    for name in ("fwd", "inv"):
        assert out["default"][name + "_exact"] and out["integer_only"][name + "_exact"], name
        assert out["default"][name + "_ms"] <= 1.2 * out["integer_only"][name + "_ms"], (name, out)   # (round 3: ~2x; the in-kernel fallback alone: 1.3x)


def test_full_batch_roundtrip_and_linearity(hx, ctx, dev, orc):
    """BASELINE config 2: fwd+inv NTT, N=16384, batch=1024 -- size-independent properties"""
    import torch
    n, batch = 16384, 1024
    q = orc.primes(1, 52, n)[0]
    t = orc.HexlTables(n, q)
    x = np.stack([orc.splitmix(n, 1000 + b, q) for b in range(batch)])
    d = _dev(hx, x, dev)
    tabs = [_dev(hx, a, dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]
    ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
    ctx.sync()
    f = hx.to_u64(d).reshape(batch, n)
    assert np.array_equal(f[0], orc.ntt_fwd(x[0], t)[0]) and np.array_equal(f[-1], orc.ntt_fwd(x[-1], t)[0])
    assert int(f.max()) < q
    # linearity: NTT(a) + NTT(b) == NTT(a + b) mod q
    s = (x[0].astype(object) + x[1].astype(object)) % q
    fs = run_fwd(hx, ctx, dev, np.array(s, dtype=np.uint64), t)[0]
    assert np.array_equal((f[0].astype(object) + f[1].astype(object)) % q, fs.astype(object))
    ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n)
    ctx.sync()
    assert torch.equal(d.cpu().reshape(-1), hx.as_i64(x).reshape(-1))


@pytest.mark.parametrize("bits", [51, 30, 55])
def test_n32768_beyond_the_reference_envelope(hx, ctx, dev, orc, bits):
    """N = 32768: 51/30-bit primes take the exact FP64 path (round 5: two 16384-point sub-transforms per polynomial,
    ntt.hip k_ntt_fwd_h / k_ntt_inv_h), 55-bit the integer butterflies (the polynomial does not fit the CU's LDS: every re-deal runs in
    two half-size rounds, ntt_core.hpp redeal_half); ALL_MAX data forces the per-polynomial integer fallback (k_ntt_redo_*)."""
    n = 32768
    q = orc.primes(1, bits, n)[0]
    tb = orc.HexlTables(n, q)
    xs = np.stack([stimulus(kind, n, q, seed=3) for kind in ("RANDOM", "RAMP", "ALL_MAX_VALUES", "IMPULSE", "RANDOM")])
    d = hx.as_i64(xs).to(dev)
    tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
    ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
    ctx.sync()
    assert np.array_equal(hx.to_u64(d), orc.ntt_fwd(xs, tb))
    ys = np.stack([stimulus("RANDOM", n, q, seed=9 + i) for i in range(4)])
    d = hx.as_i64(ys).to(dev)
    ctx.ntt_inv(d, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, n)
    ctx.sync()
    assert np.array_equal(hx.to_u64(d), orc.ntt_inv(ys, tb))


def _largest_prime_below(orc, top, n):
    v = ((top - 1) // (2 * n)) * (2 * n) + 1
    while not orc.orc().orc_is_prime(v):
        v -= 2 * n
    return v


@pytest.mark.parametrize("tier", ["lazy_51bit", "lazy_30bit", "strict_largest_below_2p52", "strict_2p52_plus_393217"])
def test_n32768_as_two_sub_transforms(hx, ctx, dev, orc, tier):
    """N = 32768 on the persistent half-transform kernels (ntt.hip k_ntt_fwd_h / k_ntt_inv_h, round 5): a radix-2 step across the halves
    plus two 16384-point sub-transforms per polynomial. A batch that makes every workgroup walk several polynomials, in both lazy and
    both strict arithmetic shapes; every fifth polynomial carries one word outside the fast path's range -- in the lower or the upper
    half -- and takes the integer fallback inside the same launch; random (non-Shoup) tables take the integer butterflies for the
    whole batch; all bit-exact against the oracle's op-for-op replay, plus the round trip at full batch."""
    n = 32768
    q = {"lazy_51bit": lambda: orc.primes(1, 51, n)[0], "lazy_30bit": lambda: orc.primes(1, 30, n)[0],
         "strict_largest_below_2p52": lambda: _largest_prime_below(orc, 1 << 52, n),
         "strict_2p52_plus_393217": lambda: 4503599627763713}[tier]()
    assert orc.orc().orc_is_prime(q) and q % (2 * n) == 1
    t = orc.HexlTables(n, q)
    batch = 700
    base = np.stack([orc.splitmix(n, 170 + b, q) for b in range(4)])
    base[3, :4] = np.array([q - 1, 0, 1, q // 2], dtype=np.uint64)
    base[3, n // 2:n // 2 + 4] = np.array([q - 1, q - 2, 0, q // 2 + 1], dtype=np.uint64)
    x = base[np.arange(batch) % 4].copy()
    odd = np.arange(2, batch, 5)
    x[odd, (odd * 40503) % n] = np.uint64((1 << 63) + 11) + odd.astype(np.uint64)
    assert ((odd * 40503) % n < n // 2).any() and ((odd * 40503) % n >= n // 2).any()
    for fwd in (True, False):
        got = (run_fwd if fwd else run_inv)(hx, ctx, dev, x, t)
        ref4 = (orc.ntt_fwd if fwd else orc.ntt_inv)(base, t)
        clean = np.setdiff1d(np.arange(batch), odd)
        assert (got[clean] == ref4[clean % 4]).all(), "canonical polynomials"
        sample = odd[:: max(1, len(odd) // 8)]
        want = (orc.ntt_fwd if fwd else orc.ntt_inv)(x[sample], t)
        assert np.array_equal(got[sample], want), "polynomials with out-of-range words"
    y = run_inv(hx, ctx, dev, run_fwd(hx, ctx, dev, x[clean], t), t)
    assert np.array_equal(y, x[clean])
    if tier == "lazy_30bit":
        rnd = orc.HexlTables(n, q)
        rng = np.random.default_rng(5)
        for arr in (rnd.roots, rnd.precon, rnd.inv_roots, rnd.inv_precon):
            arr[:] = rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)
        small = x[:300]
        assert np.array_equal(run_fwd(hx, ctx, dev, small, rnd)[::37], orc.ntt_fwd(small[::37], rnd))
        assert np.array_equal(run_inv(hx, ctx, dev, small, rnd)[::37], orc.ntt_inv(small[::37], rnd))


@pytest.mark.parametrize("env", [{"HEXL_NTT_E16": "7"}, {"HEXL_NTT_E16": "0"}, {"HEXL_NTT_INT": "1"}, {"HEXL_NTT_PERSIST": "0"}, {"HEXL_NTT_HALVES": "0"}],
                         ids=["fp64_16_per_thread", "fp64_32_per_thread", "integer_butterflies", "one_workgroup_per_polynomial", "n32768_half_size_exchanges"])
def test_alternative_geometries_agree_with_the_oracle(env):
    """the kernel variants the defaults do not select at some ring dimension (ntt.hip small_e16, HEXL_NTT_INT,
    HEXL_NTT_PERSIST; knobs are read once per process, hence a child process): N = 2048 .. 16384, batches large
    enough for the persistent kernels, forward and inverse against the oracle"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch, hexl_fpga_amd as hx, orc
dev = torch.device("cuda:0"); ctx = hx.Context(0)
ok = True
import os
for n in ((32768,) if os.environ.get("HEXL_NTT_HALVES") else (2048, 4096, 8192, 16384)):
    q = orc.primes(2, 51, n)[1]
    t = orc.HexlTables(n, q)
    x = np.stack([orc.splitmix(n, 7 + b, q) for b in range(4)])
    batch = 3000 * 2048 // n + 5
    tabs = [hx.as_i64(a).to(dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]
    for fwd in (True, False):
        d = hx.as_i64(x).to(dev).repeat((batch + 3) // 4, 1)[:batch].contiguous()
        if fwd: ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
        else: ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n)
        ctx.sync()
        got = hx.to_u64(d).reshape(batch, n)
        want = orc.ntt_fwd(x, t) if fwd else orc.ntt_inv(x, t)
        ok = ok and all((got[j::4] == want[j]).all() for j in range(4))
print("OK" if ok else "MISMATCH")
''' % (str(root), str(root / "oracle"), str(root / "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    print(out.stdout[-500:], out.stderr[-1500:])
    assert out.returncode == 0 and out.stdout.strip().endswith("OK")


@pytest.mark.parametrize("n", [1024, 4096, 16384])
def test_out_of_range_polynomials_inside_a_persistent_batch(hx, ctx, dev, orc, n):
    """a batch large enough for the persistent FP64 kernels in which every seventh polynomial carries ONE word outside the
    Harvey input range (at a different position each time): exactly those take the integer fallback (ntt.hip RangeVote),
    their neighbours the FP64 path, and all of them match the oracle's op-for-op replay"""
    q = orc.primes(1, 51, n)[0]
    t = orc.HexlTables(n, q)
    batch = 5000 * 1024 // n + 3
    base = np.stack([orc.splitmix(n, 50 + b, q) for b in range(3)])
    x = base[np.arange(batch) % 3].copy()
    bad = np.arange(3, batch, 7)
    pos = (bad * 2654435761) % n
    x[bad, pos] = np.uint64((1 << 63) + 5) + bad.astype(np.uint64)
    distinct = {}
    for fwd in (True, False):
        got = (run_fwd if fwd else run_inv)(hx, ctx, dev, x, t)
        ref3 = (orc.ntt_fwd if fwd else orc.ntt_inv)(base, t)
        clean = np.setdiff1d(np.arange(batch), bad)
        assert (got[clean] == ref3[clean % 3]).all()
        sample = bad[:: max(1, len(bad) // 24)]                    # the oracle is slow: a spread of the dirty ones
        want = (orc.ntt_fwd if fwd else orc.ntt_inv)(x[sample], t)
        assert np.array_equal(got[sample], want)
        distinct[fwd] = got


@pytest.mark.parametrize("n,batch,strict", [(16384, 1024, False), (4096, 3000, False), (16384, 700, True)],
                         ids=["16384-1024", "4096-3000", "16384-700-strict_tier_q_2p52_plus_393217"])
def test_tables_edited_in_place_between_calls(hx, ctx, dev, orc, n, batch, strict):
    """nothing about the caller's tables may be remembered across calls unless it is re-verified: the same table tensors are
    (1) reused over several calls, (2) edited IN PLACE by one word between calls (same pointers; the reference's answer for
    improper tables is the integer butterflies on exactly those words), (3) restored -- every call must match the oracle for
    the tables as they are AT THAT CALL; forward and inverse tables alternate, as in bench.py. (Written for a table cache whose
    kernels re-hash the tables -- tools/experiments/ntt_table_cache.patch, sound but no faster -- and kept as the guard for any
    future one.) The strict-tier case also covers the inverse kernel whose integer fallback is a launch of its own (ntt.hip ntt_inv_redo,
    k_ntt_redo_inv: the whole batch goes there while the tables are not Shoup tables)."""
    import torch
    q = 4503599627763713 if strict else orc.primes(1, 51, n)[0]
    t = orc.HexlTables(n, q)
    base = np.stack([orc.splitmix(n, 70 + b, q) for b in range(3)])
    x = base[np.arange(batch) % 3].copy()
    tabs = [_dev(hx, a, dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]

    def check(edit=None):
        tt = t
        if edit is not None:                                     # the oracle with the same word flipped
            import copy
            tt = copy.copy(t)
            tt.roots = t.roots.copy(); tt.inv_roots = t.inv_roots.copy()
            tt.roots[edit] ^= np.uint64(1); tt.inv_roots[edit] ^= np.uint64(1)
        for fwd in (True, False):
            d = _dev(hx, x, dev)
            if fwd:
                ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
            else:
                ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n)
            ctx.sync()
            got = hx.to_u64(d).reshape(batch, n)
            want = (orc.ntt_fwd if fwd else orc.ntt_inv)(base, tt)
            assert np.array_equal(got[:3], want) and np.array_equal(got[batch - 3:], want[(np.arange(batch - 3, batch)) % 3]), (fwd, edit)

    check(); check(); check()                                    # miss, then hits
    for pos in (5, n - 1):
        one = torch.tensor(1, dtype=torch.int64, device=dev)
        tabs[0][pos] ^= one; tabs[2][pos] ^= one                 # in place, same pointers
        check(edit=pos); check(edit=pos)                         # stale cache detected by the kernel, then verified again
        tabs[0][pos] ^= one; tabs[2][pos] ^= one
        check(); check()


def _wide_primes(orc, n):
    """SURVEY 8d's q = 2^52 + 393217 and the largest prime = 1 mod 2^15 below 2^52 * 1.125 (f64_arith.hpp STRICT_NTT_MAX_Q)"""
    top = (1 << 52) + (1 << 49)
    v = ((top - 1) // 32768) * 32768 + 1
    while not orc.orc().orc_is_prime(v):
        v -= 32768
    return [4503599627763713, v]


@pytest.mark.parametrize("which", [0, 1], ids=["q_2p52_plus_393217", "largest_below_2p52_x_1p125"])
@pytest.mark.parametrize("n", [16384, 4096])
def test_moduli_just_above_2p52_take_the_strict_fp64_path(hx, ctx, dev, orc, which, n):
    """Moduli in [2^52, 2^52 * 1.125) -- SURVEY 8d pins q = 4503599627763713 for configs 1-2 -- run the STRICT FP64 transforms
    (ntt.hip, f64_arith.hpp STRICT_NTT_MAX_Q; round 4) instead of the integer Harvey kernels: the reference's stimulus matrix,
    canonical random batches large enough for the persistent kernels, words in [q, 4q) below and above 2^53 (the latter take
    the integer fallback inside the same launch), all bit-exact against the oracle's op-for-op replay"""
    q = _wide_primes(orc, n)[which]
    assert orc.orc().orc_is_prime(q) and q % (2 * n) == 1 and (1 << 52) <= q < (1 << 52) + (1 << 49)
    t = orc.HexlTables(n, q)
    x = np.stack([stimulus(k, n, q) for k in STIMS])
    assert np.array_equal(run_fwd(hx, ctx, dev, x, t), orc.ntt_fwd(x, t))
    assert np.array_equal(run_inv(hx, ctx, dev, x, t), orc.ntt_inv(x, t))
    # a persistent-kernel batch of canonical polynomials; every fifth one carries a word in [q, 2^53) (still the fast path,
    # forward) or in [2^53, 4q) (integer fallback), every eleventh the largest canonical words
    batch = 3000 * 1024 // n + 5
    base = np.stack([orc.splitmix(n, 70 + b, q) for b in range(4)])
    base[3, :8] = np.array([q - 1, q - 2, (1 << 52), (1 << 52) + 1, (1 << 52) - 1, 0, 1, q // 2], dtype=np.uint64)
    x = base[np.arange(batch) % 4].copy()
    odd = np.arange(2, batch, 5)
    x[odd, (odd * 40503) % n] = np.where(odd % 2 == 0, np.uint64((1 << 53) - 7), np.uint64(min(4 * q - 3, (1 << 63) + 11)))
    for fwd in (True, False):
        got = (run_fwd if fwd else run_inv)(hx, ctx, dev, x, t)
        ref4 = (orc.ntt_fwd if fwd else orc.ntt_inv)(base, t)
        clean = np.setdiff1d(np.arange(batch), odd)
        assert (got[clean] == ref4[clean % 4]).all(), "canonical polynomials"
        sample = odd[:: max(1, len(odd) // 16)]
        want = (orc.ntt_fwd if fwd else orc.ntt_inv)(x[sample], t)
        assert np.array_equal(got[sample], want), "polynomials with out-of-range words"
    # round trip at full batch
    y = run_inv(hx, ctx, dev, run_fwd(hx, ctx, dev, base, t), t)
    assert np.array_equal(y, base)


def test_violation_counter_ring_wraps_and_tables_change_between_launches():
    """The persistent fast path (batch > 256) over 70 launches in one process -- the ring of 64 violation counters k_ntt_prepare fills wraps,
    and the slot a launch uses was zeroed by the launch 32 before it --, then with the SAME device arrays holding another root's tables
    (every launch re-derives its double tables), then with tables that are not Shoup tables (whole batch through the integer butterflies).
    (Round 5 ran this with HEXL_NTT_FUSED_PREPARE=1; that knob measured no faster and is archived under tools/experiments/.)"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys
sys.path[:0] = [%r, %r]
import numpy as np, torch, hexl_fpga_amd as hx, orc
dev = torch.device("cuda:0"); ctx = hx.Context(0)
n, batch = 16384, 520
rng = np.random.default_rng(11)
for bits in (51, 30):
    q = orc.primes(1, bits, n)[0]
    t = orc.HexlTables(n, q)
    x = rng.integers(0, q, size=(8, n), dtype=np.uint64)
    d_in = hx.as_i64(np.tile(x, (batch // 8, 1)).reshape(-1)).to(dev)
    tabs = [hx.as_i64(a).to(dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]
    for rep in range(70):
        d = d_in.clone()
        ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
        if rep %% 23 == 0:
            ctx.sync(); got = hx.to_u64(d).reshape(batch, n)
            assert np.array_equal(got[:8], orc.ntt_fwd(x, t)) and np.array_equal(got[-8:], orc.ntt_fwd(x, t)), ("fwd", bits, rep)
        ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n)
        if rep %% 23 == 0:
            ctx.sync(); assert np.array_equal(hx.to_u64(d).reshape(batch, n)[-8:], x), ("inv", bits, rep)
    # the same device arrays now hold ANOTHER root's tables: every launch re-derives its copies
    q2 = orc.primes(2, bits, n)[1]
    t2 = orc.HexlTables(n, q2)
    for dst, src in zip(tabs, (t2.roots, t2.precon, t2.inv_roots, t2.inv_precon)):
        dst.copy_(hx.as_i64(src).to(dev))
    x2 = x %% q2
    d = hx.as_i64(np.tile(x2, (batch // 8, 1)).reshape(-1)).to(dev)
    ctx.ntt_fwd(d, tabs[0], tabs[1], q2, n); ctx.sync()
    assert np.array_equal(hx.to_u64(d).reshape(batch, n)[-8:], orc.ntt_fwd(x2, t2)), ("edited tables", bits)
    # not Shoup tables at all
    t2.roots[:] = rng.integers(0, q2, size=n, dtype=np.uint64); t2.precon[:] = rng.integers(0, q2, size=n, dtype=np.uint64)
    tabs[0].copy_(hx.as_i64(t2.roots).to(dev)); tabs[1].copy_(hx.as_i64(t2.precon).to(dev))
    d = hx.as_i64(np.tile(x2, (batch // 8, 1)).reshape(-1)).to(dev)
    ctx.ntt_fwd(d, tabs[0], tabs[1], q2, n); ctx.sync()
    assert np.array_equal(hx.to_u64(d).reshape(batch, n)[-8:], orc.ntt_fwd(x2, t2)), ("random tables", bits)
print("OK")
''' % (str(root), str(root / "oracle"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    print(out.stdout[-500:], out.stderr[-1500:])
    assert out.returncode == 0 and out.stdout.strip().endswith("OK")
