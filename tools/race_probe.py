#!/usr/bin/env python3
"""race_probe.py [n] [L] [K] [instances] [launches] -- the inverse-after-inverse LDS race found by tools/soak_ks_random.py (round 6): a
keyswitch of many small workgroups per CU, repeated; every instance of every launch against the oracle. Before the readers' gate
(ntt_core.hpp) a few instances in 10^4 came back with a wrong k = 0 half at n = 2048; the count must be 0.
HEXL_MI355X_LIB selects another build of the library, the HEXL_* knobs apply as usual."""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc
from ks_util import KsCase

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else L + 1
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 4500
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
ctx = hx.Context(0)
moduli = {(2048, 3): [4503599611383809, 2269392289533953, 38677947719681]}.get((n, K))
case = KsCase(orc, n, L, K, seed=5, moduli=moduli)
plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
plan.set_keys(case.keys)
ins = [case.inputs(orc, b) for b in range(3)]
want = torch.from_numpy(np.stack([case.expected(orc, t, r) for t, r in ins]).view(np.int64)).to(dev)
d_t = hx.as_i64(np.concatenate([ins[b % 3][0] for b in range(nb)])).to(dev)
d_r0 = hx.as_i64(np.concatenate([ins[b % 3][1] for b in range(nb)])).to(dev)
idx = torch.arange(nb, device=dev) % 3
bad = halves = 0
for _ in range(reps):
    d_r = d_r0.clone()
    plan.keyswitch(d_r, d_t, nb)
    ctx.sync()
    diff = (d_r.view(nb, 2, -1) != want[idx].view(nb, 2, -1)).any(dim=2)
    bad += int(diff.any(dim=1).sum())
    halves += int(diff[:, 0].sum()) * 10 + int(diff[:, 1].sum())
env = {k: v for k, v in os.environ.items() if k.startswith("HEXL_")}
print(f"n={n} L={L} K={K} tiers={plan.tiers()[0]} {nb} instances x {reps} launches {env}: {bad} wrong instances "
      f"(k = 0 halves: {halves // 10}, k = 1 halves: {halves % 10})")
sys.exit(1 if bad else 0)
