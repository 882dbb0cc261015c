#!/usr/bin/env python3
"""soak.py [iterations] -- determinism soak: the same keyswitch / NTT launches repeated many times must give
bit-identical outputs every time (a data race in the barrier-light LDS re-deals would show up as a rare mismatch)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc
import bench
from ks_util import KsCase

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
ctx = hx.Context(0)
bad = 0
# (the last two shapes take the quarter-transform latency path, keyswitch_lat.hip: one and three instances, integer atomics)
for (L, K, B) in ((7, 8, 512), (6, 7, 300), (3, 4, 40), (6, 7, 1), (6, 7, 3), (7, 8, 2)):
    case = KsCase(orc, 16384, L, K, seed=L)
    plan = hx.KeySwitchPlan(ctx, 16384, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    d_t, d_r0 = bench.device_inputs(hx, orc, case, B, dev)
    ref = None
    for it in range(iters):
        d_r = d_r0.clone()
        plan.keyswitch(d_r, d_t, B)
        ctx.sync()
        if ref is None:
            ref = d_r.clone()
        elif not torch.equal(ref, d_r):
            bad += 1
            print(f"MISMATCH keyswitch L={L} batch={B} iteration {it}: {(ref != d_r).sum().item()} words differ")
    plan.close()
    print(f"keyswitch L={L} K={K} batch={B}: {iters} identical runs" if not bad else "see above")
# forward then inverse NTT, batch 1024: every run identical, and the round trip is the identity
N = 16384
q = orc.primes(1, 51, N)[0]
tb = orc.HexlTables(N, q)
tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
x0 = hx.as_i64(np.stack([orc.splitmix(N, 1000 + b, q) for b in range(1024)])).to(dev).contiguous()
ref = None
for it in range(iters):
    x = x0.clone()
    ctx.ntt_fwd(x, tabs[0], tabs[1], q, N)
    ctx.sync()
    if ref is None:
        ref = x.clone()
    elif not torch.equal(ref, x):
        bad += 1
        print(f"MISMATCH forward NTT iteration {it}")
    ctx.ntt_inv(x, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, N)
    ctx.sync()
    if not torch.equal(x, x0):
        bad += 1
        print(f"MISMATCH NTT round trip iteration {it}")
print(f"NTT batch 1024: {iters} identical forward runs and identity round trips" if not bad else "see above")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
