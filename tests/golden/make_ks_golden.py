#!/usr/bin/env python3
"""Regenerates tests/golden/ks_golden.json: FNV-1a-64 digests (+ three sample words) of the keyswitch oracle's output
for the BASELINE shapes, on the synthetic instances of tests/ks_util.py (moduli =
GeneratePrimes(K, 51, 16384), everything uniform mod its limb from splitmix64 seeds).

    python tests/golden/make_ks_golden.py

These vectors are produced by THIS repository's oracle (oracle/hexl_oracle.c::orc_keyswitch), not by the reference: the
reference's keyswitch vectors are an external download (README.md:166-176) and its SYCL kernels cannot be built here, so
keyswitch parity stays "unpinned against reference vectors" (DESIGN.md 2). What the fixture pins is (a) the oracle
against silent change -- the independent big-integer models and the RLWE checks of tests/test_oracle_keyswitch.py
validated exactly these bits -- and (b) the GPU output of the same instances (tests/test_gpu_keyswitch.py)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np  # noqa: E402
import orc  # noqa: E402
from ks_util import KsCase  # noqa: E402

SHAPES = [(16384, 6, 7, 51), (16384, 7, 8, 51), (16384, 6, 7, 48), (8192, 5, 7, 51), (1024, 1, 2, 51)]


def main():
    orc.build(with_ref=False)
    out = {"format": "fnv1a64 over the little-endian bytes of result[2][L][n] after orc_keyswitch; seed = n + L",
           "vectors": []}
    for n, L, K, bits in SHAPES:
        case = KsCase(orc, n, L, K, seed=n + L, bits=bits)
        for b in range(2):
            t, r = case.inputs(orc, b)
            e = case.expected(orc, t, r)
            out["vectors"].append({"n": n, "L": L, "K": K, "bits": bits, "instance": b,
                                   "moduli": [int(v) for v in case.moduli],
                                   "fnv_t_target": f"{orc.fnv(t):016x}", "fnv_result_in": f"{orc.fnv(r):016x}",
                                   "fnv_result_out": f"{orc.fnv(e):016x}",
                                   "out_first_mid_last": [int(e[0]), int(e[len(e) // 2]), int(e[-1])]})
    (Path(__file__).parent / "ks_golden.json").write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", len(out["vectors"]), "vectors")


if __name__ == "__main__":
    main()
