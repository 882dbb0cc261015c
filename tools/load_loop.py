#!/usr/bin/env python3
"""One kernel family in a loop for `seconds` (for tools/power_probe.sh): load_loop.py ntt|inv|dyadic|ks [seconds]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc
import bench
from ks_util import KsCase

kind = sys.argv[1] if len(sys.argv) > 1 else "ntt"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
ctx = hx.Context(0)
N = 16384
if kind in ("ntt", "inv"):
    q = orc.primes(1, 51, N)[0]
    tb = orc.HexlTables(N, q)
    x = hx.as_i64(np.stack([orc.splitmix(N, 1000 + b, q) for b in range(8)])).to(dev).repeat(512, 1).contiguous()
    tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
    run = (lambda: ctx.ntt_fwd(x, tabs[0], tabs[1], q, N)) if kind == "ntt" else (lambda: ctx.ntt_inv(x, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, N))
    unit = 4096
elif kind == "dyadic":
    n, nm, batch = 8192, 4, 4096
    mod1 = np.array(orc.primes(nm, 52, n), dtype=np.uint64)
    one = np.concatenate([orc.splitmix(n, 3 + i, int(m)) for _ in range(2) for i, m in enumerate(mod1)])
    a = hx.as_i64(one).to(dev).repeat(batch)
    mod = hx.as_i64(np.tile(mod1, batch)).to(dev)
    out = torch.empty(batch * 3 * nm * n, dtype=torch.int64, device=dev)
    run = lambda: ctx.dyadic_multiply(out, a, a, mod, n, nm)
    unit = batch
else:
    case = KsCase(orc, N, 7, 8, seed=1)
    plan = hx.KeySwitchPlan(ctx, N, 7, 8, 8, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    d_t, d_r = bench.device_inputs(hx, orc, case, 4096, dev)
    run = lambda: plan.keyswitch(d_r, d_t, 4096)
    unit = 4096
run(); torch.cuda.synchronize()
print("START", flush=True)
t0, done = time.perf_counter(), 0
while time.perf_counter() - t0 < seconds:
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    done += 20 * unit
print(f"{kind}: {done / (time.perf_counter() - t0):,.0f} units/s over {seconds:.0f} s")
