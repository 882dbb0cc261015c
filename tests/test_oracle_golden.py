"""CPU: pins the oracle (oracle/hexl_oracle.c) to the reference.
 - committed golden fixtures (tests/golden/ntt_golden.json, generated from the reference's own CPU oracle
   by tests/golden/make_golden.py);
 - when oracle/_ref is present (build container), direct comparison with the reference code on random data."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = json.loads((Path(__file__).parent / "golden" / "ntt_golden.json").read_text())["records"]


@pytest.mark.parametrize("rec", GOLD, ids=lambda r: f"n{r['n']}_b{r['bits']}")
def test_oracle_matches_golden(orc, rec):
    n, q = rec["n"], rec["q"]
    assert orc.primes(1, rec["bits"], n)[0] == q
    t = orc.HexlTables(n, q)
    assert (t.w, t.inv_n, t.inv_n_w) == (rec["w"], rec["inv_n"], rec["inv_n_w"])
    for arr, key in ((t.roots, "fnv_roots"), (t.precon, "fnv_precon"), (t.inv_roots, "fnv_inv_roots"),
                     (t.inv_precon, "fnv_inv_precon")):
        assert "%016x" % orc.fnv(arr) == rec[key], key
    ks = np.zeros(4 * n, dtype=np.uint64)
    orc.orc().orc_tables_keyswitch(n, q, t.w, orc.p(ks))
    assert "%016x" % orc.fnv(ks) == rec["fnv_ks_block"]
    for name, s in rec["stimuli"].items():
        x = {"RAMP": np.arange(n, dtype=np.uint64), "ALLMAX": np.full(n, 2**64 - 1, dtype=np.uint64),
             "SPLITMIX42": orc.splitmix(n, 42, q)}[name]
        f, i = orc.ntt_fwd(x, t)[0], orc.ntt_inv(x, t)[0]
        assert "%016x" % orc.fnv(f) == s["fwd_fnv"] and [int(v) for v in f[:4]] == s["fwd_head"]
        assert "%016x" % orc.fnv(i) == s["inv_fnv"] and [int(v) for v in i[-4:]] == s["inv_tail"]


def test_survey_sample_values(orc):
    """SURVEY 8c table, 52-bit row (BASELINE config 1): q, w, inv_n, inv_n_w and output samples"""
    n = 16384
    q = orc.primes(1, 52, n)[0]
    t = orc.HexlTables(n, q)
    assert (q, t.w, t.inv_n, t.inv_n_w) == (4503599627763713, 51902047037, 4503324749856745, 133753238635015)
    f = orc.ntt_fwd(np.arange(n, dtype=np.uint64), t)[0]
    assert [int(f[0]), int(f[1]), int(f[-1])] == [1661452251559784, 898320242551355, 1451786022035036]
    i = orc.ntt_inv(orc.splitmix(n, 42, q), t)[0]
    assert [int(i[0]), int(i[1]), int(i[-1])] == [3863749938896052, 3089095866671124, 4398034571517011]
    assert orc.primes(8, 51, n) == [2251799814045697, 2251799814799361, 2251799814930433, 2251799815094273,
                                    2251799815487489, 2251799815520257, 2251799816273921, 2251799816568833]


def test_against_reference_library(orc):
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent on this box)")
    rng = np.random.default_rng(1)
    for n, bits in ((16384, 62), (16384, 20), (4096, 40), (1024, 30)):
        q = orc.primes(1, bits, n)[0]
        t = orc.HexlTables(n, q)
        assert t.w == R.ref_minimal_primitive_root(2 * n, q)
        # random (non-root) tables + out-of-range data through the free functions
        tabs = [rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64) for _ in range(4)]
        x = rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)
        a, b = x.copy(), x.copy()
        orc.orc().orc_ntt_fwd(orc.p(a), n, q, orc.p(tabs[0]), orc.p(tabs[1]))
        R.ref_fwd_with_tables(orc.p(b), n, q, orc.p(tabs[0]), orc.p(tabs[1]))
        assert np.array_equal(a, b)
        # inverse: the reference free function derives inv_n/inv_n_w from its table, mirror that
        tabs[2] %= q
        a, b = x.copy(), x.copy()
        inv_n = orc.orc().orc_invmod(n, q)
        inv_n_w = orc.orc().orc_mulmod(inv_n, int(tabs[2][n - 1]), q)
        orc.orc().orc_ntt_inv(orc.p(a), n, q, orc.p(tabs[2]), orc.p(tabs[3]), inv_n, inv_n_w)
        R.ref_inv_with_tables(orc.p(b), n, q, orc.p(tabs[2]), orc.p(tabs[3]))
        assert np.array_equal(a, b)
