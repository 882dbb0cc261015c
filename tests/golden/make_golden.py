"""Regenerates tests/golden/ntt_golden.json from the REFERENCE's own CPU code
(oracle/_ref/libhexlfpga_ref.so = /root/reference/tests/test_utils/ntt.cpp +
host/src/{number_theory_util,twiddle-factors}.cpp, built by `make -C oracle ref`).
Run only in the build container (needs /root/reference):  python tests/golden/make_golden.py

Each record pins, for N and prime q = GeneratePrimes(1, bits, N)[0]:
  w = MinimalPrimitiveRoot(2N, q), inv_n, inv_n_w, FNV-1a-64 digests of the four HEXL-layout
  tables and of the hexl-fpga keyswitch twiddle block, and for the stimuli RAMP / ALLMAX /
  SPLITMIX42 the digest + first/last 4 words of ComputeForward/ComputeInverse(out, in, 1, 1).
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import orc  # noqa: E402

orc.build(with_ref=True)
R = orc.ref()
assert R is not None, "reference library not built (needs /root/reference)"
p = orc.p


def rec(n, bits):
    pr = np.zeros(1, dtype=np.uint64)
    assert R.ref_generate_primes(p(pr), 1, bits, n) == 1
    q = int(pr[0])
    t = [np.zeros(n, dtype=np.uint64) for _ in range(4)]
    w = R.ref_ntt_tables(n, q, *[p(x) for x in t])
    inv_n = R.ref_inverse_mod(n, q)
    inv_n_w = int((int(inv_n) * int(t[2][n - 1])) % q)
    ks = np.zeros(4 * n, dtype=np.uint64)
    R.ref_ks_tables(n, q, p(ks))
    out = {"n": n, "bits": bits, "q": q, "w": int(w), "inv_n": int(inv_n), "inv_n_w": inv_n_w,
           "fnv_roots": "%016x" % orc.fnv(t[0]), "fnv_precon": "%016x" % orc.fnv(t[1]),
           "fnv_inv_roots": "%016x" % orc.fnv(t[2]), "fnv_inv_precon": "%016x" % orc.fnv(t[3]),
           "fnv_ks_block": "%016x" % orc.fnv(ks), "stimuli": {}}
    for name in ("RAMP", "ALLMAX", "SPLITMIX42"):
        if name == "RAMP":
            x = np.arange(n, dtype=np.uint64)
        elif name == "ALLMAX":
            x = np.full(n, 2**64 - 1, dtype=np.uint64)
        else:
            x = orc.splitmix(n, 42, q)
        f = np.zeros(n, dtype=np.uint64)
        i = np.zeros(n, dtype=np.uint64)
        R.ref_ntt_forward(p(f), p(x), n, q)
        R.ref_ntt_inverse(p(i), p(x), n, q)
        out["stimuli"][name] = {
            "fwd_fnv": "%016x" % orc.fnv(f), "fwd_head": [int(v) for v in f[:4]], "fwd_tail": [int(v) for v in f[-4:]],
            "inv_fnv": "%016x" % orc.fnv(i), "inv_head": [int(v) for v in i[:4]], "inv_tail": [int(v) for v in i[-4:]],
        }
    return out


records = [rec(16384, b) for b in (20, 32, 51, 52, 55, 62)]
records += [rec(n, 51) for n in (1024, 2048, 4096, 8192)]
dst = Path(__file__).with_name("ntt_golden.json")
dst.write_text(json.dumps({"source": "intel/hexl-fpga v2.0 tests/test_utils/ntt.cpp via oracle/_ref",
                           "digest": "FNV-1a-64 over little-endian bytes", "records": records}, indent=1))
print("wrote", dst, len(records), "records")
