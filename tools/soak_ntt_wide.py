#!/usr/bin/env python3
"""Randomised parity soak of the standalone NTT at moduli in [2^52, 2^52 x 1.125) (strict FP64 kernels, ntt.hip) against the oracle:
random primes = 1 mod 2n, every ring dimension, single-polynomial and persistent-kernel batches, canonical words with the extremes
mixed in and a few out-of-range words. usage: soak_ntt_wide.py [seconds = 120]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda:0")
ctx = hx.Context(0)
rng = np.random.default_rng(20260929)
LO, HI = 1 << 52, (1 << 52) + (1 << 49)
t0 = time.time()
cases = fails = 0
while time.time() - t0 < budget:
    n = int(rng.choice([1024, 2048, 4096, 8192, 16384, 32768]))
    v = int(rng.integers(LO, HI)) // (2 * n) * (2 * n) + 1
    while v >= HI or v < LO or not orc.orc().orc_is_prime(v):
        v -= 2 * n
        if v < LO:
            v = HI - 1 - (HI - 2) % (2 * n)
    q = v
    t = orc.HexlTables(n, q)
    nuniq = 6
    base = np.stack([orc.splitmix(n, int(rng.integers(1, 1 << 30)), q) for _ in range(nuniq)])
    base[0, :6] = np.array([q - 1, q - 2, 1 << 52, (1 << 52) - 1, 0, q // 2 + 1], dtype=np.uint64)
    base[1, int(rng.integers(0, n))] = np.uint64((1 << 53) - 1)          # in [q, 2^53): fast path forward, fallback inverse if >= 2q
    base[2, int(rng.integers(0, n))] = np.uint64(min(4 * q - 1, (1 << 63) + 3))   # >= 2^53: integer fallback
    batch = int(rng.choice([1, 5, 300, 3000 * 1024 // n + 7]))
    x = base[np.arange(batch) % nuniq].copy()
    tabs = [hx.as_i64(a).to(dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]
    for fwd in (True, False):
        d = hx.as_i64(x).to(dev)
        if fwd:
            ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
        else:
            ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n)
        ctx.sync()
        got = hx.to_u64(d).reshape(batch, n)
        want = (orc.ntt_fwd if fwd else orc.ntt_inv)(base, t)
        ok = all(np.array_equal(got[b], want[b % nuniq]) for b in range(min(batch, 40))) and np.array_equal(got[-1], want[(batch - 1) % nuniq])
        cases += 1
        if not ok:
            fails += 1
            print(f"MISMATCH n={n} q={q} batch={batch} fwd={fwd}")
print(f"soak_ntt_wide: {cases} cases in {time.time() - t0:.0f} s, {fails} mismatches")
sys.exit(1 if fails else 0)
