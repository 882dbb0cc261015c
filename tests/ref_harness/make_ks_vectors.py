#!/usr/bin/env python3
"""Writes keyswitch test vectors in the JSON format the reference's tests read (tests/test_keyswitch.cpp:59-103):
file name {N}_{decomp}_{key_modulus}_{rns}_{kcc}_{idx}.json with coeff_count, decomp_modulus_size, key_modulus_size,
rns_modulus_size, key_component_count, moduli, modswitch_factors, the four twiddle arrays (hexl-fpga layout,
host/src/twiddle-factors.cpp:16-62), key_vector, t_target_iter_ptr, input and expected_output.
The official vectors (testdata.zip, README.md:166-176) are an external download; these stand in for them with
expected_output computed by the CPU oracle (test infrastructure).  usage: make_ks_vectors.py OUTDIR N L K RNS COUNT"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "oracle"), str(ROOT / "tests")]
import orc  # noqa: E402
from ks_util import KsCase  # noqa: E402


def main(outdir, n, L, K, rns, count):
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    case = KsCase(orc, n, L, K, seed=n + 10 * L + K, with_twiddles=True)
    case.rns = rns
    tw = case.twiddles.reshape(K, 4, n)
    for idx in range(count):
        t, r = case.inputs(orc, idx)
        exp = case.expected(orc, t, r)
        js = {
            "coeff_count": n, "decomp_modulus_size": L, "key_modulus_size": K, "rns_modulus_size": rns,
            "key_component_count": 2,
            "moduli": [int(v) for v in case.moduli], "modswitch_factors": [int(v) for v in case.modswitch],
            "inv_root_of_unity_powers": tw[:, 0].tolist(), "precon64_inv_root_of_unity_powers": tw[:, 1].tolist(),
            "root_of_unity_powers": tw[:, 2].tolist(), "precon64_root_of_unity_powers": tw[:, 3].tolist(),
            "key_vector": [k.tolist() for k in case.keys],
            "t_target_iter_ptr": t.tolist(), "input": r.tolist(), "expected_output": exp.tolist(),
        }
        (out / f"{n}_{L}_{K}_{rns}_2_{idx}.json").write_text(json.dumps(js))


if __name__ == "__main__":
    main(sys.argv[1], *[int(v) for v in sys.argv[2:7]])
