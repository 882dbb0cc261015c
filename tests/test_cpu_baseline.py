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
