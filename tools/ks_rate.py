#!/usr/bin/env python3
"""keyswitch/s of the current library under the caller's environment (HEXL_KS_INT, HEXL_KS_PIPE, HEXL_KSI_LOGE, ...),
after checking three instances against the oracle. usage: ks_rate.py [batch] [decomp] [bits] [reps] [n]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 7
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 51
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
n = int(sys.argv[5]) if len(sys.argv) > 5 else 16384
dev = torch.device("cuda:0")
ctx = hx.Context(0)
case = KsCase(orc, n, L, L + 1, seed=1, bits=bits)
plan = hx.KeySwitchPlan(ctx, n, L, L + 1, L + 1, 2, case.moduli, case.modswitch)
plan.set_keys(case.keys)
ins = [case.inputs(orc, b) for b in range(3)]
nb = 300
d_t = hx.as_i64(np.concatenate([ins[b % 3][0] for b in range(nb)])).to(dev)
d_r = hx.as_i64(np.concatenate([ins[b % 3][1] for b in range(nb)])).to(dev)
plan.keyswitch(d_r, d_t, nb)
ctx.sync()
out = hx.to_u64(d_r).reshape(nb, -1)
want = [case.expected(orc, t, r) for t, r in ins]
ok = all(np.array_equal(out[b], want[b % 3]) for b in range(nb))
d_t, d_r = bench.device_inputs(hx, orc, case, B, dev)
for _ in range(2):
    plan.keyswitch(d_r, d_t, B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    plan.keyswitch(d_r, d_t, B)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"parity={'OK' if ok else 'MISMATCH'} batch={B} L={L} bits={bits}: {B / dt:,.0f} keyswitch/s ({dt * 1e3:.2f} ms)")
