"""CPU: the OpenMP CPU port that bench.py times as `cpu_baseline` (oracle/cpu_baseline.c: Harvey/Shoup butterflies, tables
once per parameter set, one thread per ciphertext) returns exactly what the line-by-line oracle returns."""
import numpy as np
import pytest

from ks_util import KsCase


@pytest.mark.parametrize("n,L,K,threads", [(1024, 2, 3, 1), (4096, 3, 4, 3), (16384, 6, 7, 0)])
def test_cpu_port_equals_oracle(orc, n, L, K, threads):
    case = KsCase(orc, n, L, K, seed=7)
    nb = 3
    ts, rs = zip(*[case.inputs(orc, b) for b in range(nb)])
    cb = orc.CpuKeySwitch(n, L, K, case.moduli, case.keys, case.modswitch, march="x86-64-v3")
    got = np.concatenate(rs).copy()
    used = cb.keyswitch_batch(got, np.concatenate(ts), threads)
    assert used >= 1
    cb.close()
    want = np.concatenate([case.expected(orc, t, r) for t, r in zip(ts, rs)])
    assert np.array_equal(got, want)


def test_cpu_port_forward_ntt(orc):
    n = 4096
    case = KsCase(orc, n, 1, 2, seed=3)
    cb = orc.CpuKeySwitch(n, 1, 2, case.moduli, case.keys, case.modswitch, march="x86-64-v3")
    q = int(case.moduli[0])
    x = np.stack([orc.splitmix(n, 50 + b, q) for b in range(4)])
    got = x.copy().reshape(-1)
    cb.ntt_fwd_batch(got, 0, 2)
    cb.close()
    t = orc.HexlTables(n, q)
    assert np.array_equal(got.reshape(4, n), orc.ntt_fwd(x, t))


@pytest.mark.parametrize("restrict", ["", "dq", "scalar"])
@pytest.mark.parametrize("bits,n,L,K", [(51, 16384, 3, 4), (48, 4096, 2, 3), (30, 1024, 2, 3)])
def test_vector_kernels_equal_oracle(orc, monkeypatch, restrict, bits, n, L, K):
    """the AVX-512 kernel families of the CPU baseline (IFMA with 52-bit Shoup factors below 2^50, 64-bit lanes above --
    HEXL's selection rule) built for THIS host: every path returns the oracle's words. On a host without AVX-512 all three
    runs take the scalar port (the assertion on isa() says which one ran)."""
    if restrict:
        monkeypatch.setenv("HEXL_CPU_ISA", restrict)
    else:
        monkeypatch.delenv("HEXL_CPU_ISA", raising=False)
    case = KsCase(orc, n, L, K, seed=11, bits=bits)
    cb = orc.CpuKeySwitch(n, L, K, case.moduli, case.keys, case.modswitch, march="native")
    isa = cb.isa()
    assert isa in ("scalar", "avx512dq", "avx512ifma")
    if restrict == "scalar":
        assert isa == "scalar"
    if restrict == "dq":
        assert isa != "avx512ifma"
    if isa == "avx512ifma":
        assert bits < 50
    ts, rs = zip(*[case.inputs(orc, b) for b in range(2)])
    got = np.concatenate(rs).copy()
    cb.keyswitch_batch(got, np.concatenate(ts), 2)
    cb.close()
    assert np.array_equal(got, np.concatenate([case.expected(orc, t, r) for t, r in zip(ts, rs)])), isa


def test_vector_kernels_on_a_mixed_prime_chain(orc):
    """bridge-seal's chain 52,30,30,40,27,27,27 at a lower level (5 decomposition limbs of 7 key moduli): IFMA and 64-bit
    kernels side by side in one plan, operands of the mod-up reductions wider than 52 bits' worth of the small moduli"""
    from ks_util import primes_below
    n, K, L = 2048, 7, 5
    moduli = []
    for bits in (52, 30, 30, 40, 27, 27, 27):
        moduli.append(next(p for p in primes_below(orc, 8, 1 << bits, n) if p not in moduli))
    case = KsCase(orc, n, L, K, seed=5, moduli=moduli)
    cb = orc.CpuKeySwitch(n, L, K, case.moduli, case.keys, case.modswitch, march="native")
    t, r = case.inputs(orc, 0)
    got = r.copy()
    cb.keyswitch_batch(got, t, 1)
    cb.close()
    assert np.array_equal(got, case.expected(orc, t, r))


@pytest.mark.parametrize("threads", [1, 3])
def test_timed_numa_leg_equals_oracle(orc, threads):
    """cb_keyswitch_timed (bench.py's CPU leg since round 4: pinned threads, private ciphertexts, NUMA replicas of keys and tables)
    runs the same keyswitch: thread 0's first result equals the oracle's, the counters make sense, the plan survives a second leg"""
    n, L, K = 4096, 3, 4
    case = KsCase(orc, n, L, K, seed=21)
    cb = orc.CpuKeySwitch(n, L, K, case.moduli, case.keys, case.modswitch, march="native")
    ts, rs = zip(*[case.inputs(orc, b) for b in range(2)])
    for _ in range(2):
        done, el, nodes, first = cb.keyswitch_timed(np.concatenate(ts), np.concatenate(rs), threads, 0.15)
        assert done >= threads and 0.1 < el < 5.0 and nodes >= 1
        assert np.array_equal(first, case.expected(orc, ts[0], rs[0]))
    # the plain batch entry point still works on the same plan afterwards
    got = rs[1].copy()
    cb.keyswitch_batch(got, ts[1], 1)
    assert np.array_equal(got, case.expected(orc, ts[1], rs[1]))
    cb.close()
