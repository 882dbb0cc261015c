#!/usr/bin/env python3
"""Randomised parity soak of the dyadic multiply against the oracle: random ring dimension, 1-8 moduli anywhere in [2, 2^62) -- primes,
toy non-prime values (the reference's own stimulus uses 10, 20, 30, ...), powers of two +- 1 --, operands anywhere in the 64-bit range
(the contract reduces them), random batch; every item compared. usage: soak_dyadic_random.py [seconds = 60] [seed]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
ctx = hx.Context(0)
rng = np.random.default_rng(seed)
t0 = time.time()
cases = fails = 0
while time.time() - t0 < budget:
    n = int(rng.choice([1024, 2048, 4096, 8192, 16384, 32768]))
    nm = int(rng.integers(1, 9))
    kinds = rng.integers(0, 4, nm)
    moduli = []
    for k in kinds:
        if k == 0:
            moduli.append(int(rng.integers(2, 1000)))
        elif k == 1:
            e = int(rng.integers(2, 62)); moduli.append((1 << e) + int(rng.integers(-1, 2)))
        elif k == 2:
            moduli.append(int(rng.integers(1 << 61, (1 << 62) - 1)))
        else:
            moduli.append(int(rng.integers(2, 1 << int(rng.integers(2, 62)))))
    moduli = np.array([max(2, m) for m in moduli], dtype=np.uint64)
    wide = bool(rng.integers(0, 2))
    def operand(s):
        w = np.concatenate([orc.splitmix(n, s * 64 + j, 0 if wide else int(moduli[j % nm])) for j in range(2 * nm)])
        return w
    distinct = 3
    A = [operand(int(rng.integers(1, 1 << 20))) for _ in range(distinct)]
    B = [operand(int(rng.integers(1, 1 << 20))) for _ in range(distinct)]
    want = [orc.dyadic(a, b, n, moduli, exact=True) for a, b in zip(A, B)]
    batch = int(rng.choice([1, 2, 7, 64, 300]))
    idx = torch.arange(batch, device=dev) % distinct
    d_a = torch.from_numpy(np.stack(A).view(np.int64)).to(dev)[idx].contiguous()
    d_b = torch.from_numpy(np.stack(B).view(np.int64)).to(dev)[idx].contiguous()
    d_w = torch.from_numpy(np.stack(want).view(np.int64)).to(dev)[idx]
    d_m = torch.from_numpy(np.tile(moduli, batch).view(np.int64)).to(dev)
    d_o = torch.full((batch, 3 * nm * n), -1, dtype=torch.int64, device=dev)
    ctx.dyadic_multiply(d_o, d_a, d_b, d_m, n, nm)
    ctx.sync()
    wrong = int((d_o != d_w).any(dim=1).sum())
    cases += 1
    if wrong:
        fails += 1
        print(f"MISMATCH n={n} moduli={list(map(int, moduli))} wide={wide} batch={batch}: {wrong} items", flush=True)
print(f"soak_dyadic_random: {cases} cases in {time.time() - t0:.0f} s (seed {seed}), mismatches: {fails}")
sys.exit(1 if fails else 0)
