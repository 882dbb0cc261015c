#!/usr/bin/env python3
"""Randomised parity soak of the keyswitch against the oracle: random primes = 1 mod 2n of random sizes in [2^27, 2^52) (so every
FP64 tier and every mix of tiers across the limbs of one plan comes up), random ring dimension, decomposition size and batch --
batches on both sides of the slot-major threshold, so the slot-major, (b, d)-major and latency kernels all run --, uniform and
worst-case (ks_util.extreme_words) keys and inputs. Every instance of a batch is one of three distinct ones; all are compared.
usage: soak_ks_random.py [seconds = 240] [seed]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc
from ks_util import KsCase

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260930
dev = torch.device("cuda:0")
ctx = hx.Context(0)
rng = np.random.default_rng(seed)
EDGES = [1 << 49, 1 << 50, (1 << 51) + (1 << 44), 1 << 52]          # tier tops (f64_arith.hpp): period 12 / 6 / 3 / strict


def random_prime(n, used):
    while True:
        kind = rng.integers(0, 3)
        if kind == 0:                                               # just below a tier top
            v = int(EDGES[rng.integers(0, 4)]) - int(rng.integers(1, 1 << 30))
        elif kind == 1:                                             # just above a tier top (not above 2^52)
            v = int(EDGES[rng.integers(0, 3)]) + int(rng.integers(1, 1 << 30))
        else:                                                       # any size from 27 bits up
            b = int(rng.integers(27, 53))
            v = int(rng.integers(1 << (b - 1), 1 << b))
        v = v // (2 * n) * (2 * n) + 1
        while v > (1 << 26) and (v in used or not orc.orc().orc_is_prime(v)):
            v -= 2 * n
        if v > (1 << 26) and v < (1 << 52):
            return v


t0 = time.time()
cases = fails = 0
seen_tiers = set()
while time.time() - t0 < budget:
    n = int(rng.choice([1024, 2048, 4096, 8192, 16384, 16384, 16384, 32768]))
    L = int(rng.integers(1, 8))
    K = L + 1 + int(rng.integers(0, 2)) * int(rng.integers(0, 3))
    moduli = []
    for _ in range(K):
        moduli.append(random_prime(n, moduli))
    per_cu = max(1, 16384 // n)
    nb = int(rng.choice([1, 2, 3, 5, 17, 40 * per_cu, 70 * per_cu, 300 * per_cu // max(1, L // 2)]))
    extreme = bool(rng.integers(0, 2))
    case = KsCase(orc, n, L, K, seed=int(rng.integers(1, 1 << 20)), moduli=moduli, extreme_keys=extreme)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    seen_tiers.update(plan.tiers()[0] if hasattr(plan, "tiers") else [])
    ins = [case.extreme_inputs(orc, b) if extreme else case.inputs(orc, b) for b in range(3)]
    d_t = hx.as_i64(np.concatenate([ins[b % 3][0] for b in range(nb)])).to(dev)
    d_r = hx.as_i64(np.concatenate([ins[b % 3][1] for b in range(nb)])).to(dev)
    plan.keyswitch(d_r, d_t, nb)
    ctx.sync()
    out = hx.to_u64(d_r).reshape(nb, -1)
    want = [case.expected(orc, t, r) for t, r in ins]
    ok = all(np.array_equal(out[b], want[b % 3]) for b in range(nb))
    cases += 1
    if not ok:
        fails += 1
        print(f"MISMATCH n={n} L={L} K={K} nb={nb} extreme={extreme} moduli={moduli}", flush=True)
    plan.close()
    del d_t, d_r
print(f"{cases} random keyswitch cases in {time.time() - t0:.0f} s (seed {seed}), tiers seen {sorted(seen_tiers)}, mismatches: {fails}")
sys.exit(1 if fails else 0)
