#!/usr/bin/env python3
"""keyswitch/s at N = 16384 with the LARGEST 52-bit primes = 1 mod 2N (what SEAL's CoeffModulus::Create(n, {52, ...}) picks): above the
lazy bound 2^51 (1 + 2^-7), i.e. the strict FP64 kernels. usage: ks_rate_top52.py [batch = 4096] [L = 7] [reps = 20]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 16384
mods, v = [], (1 << 52) - 32767
while len(mods) < L + 1:
    if orc.orc().orc_is_prime(v):
        mods.append(v)
    v -= 32768
dev = torch.device("cuda:0")
ctx = hx.Context(0)
case = KsCase(orc, n, L, L + 1, seed=1, moduli=mods)
plan = hx.KeySwitchPlan(ctx, n, L, L + 1, L + 1, 2, case.moduli, case.modswitch)
plan.set_keys(case.keys)
ins = [case.inputs(orc, b) for b in range(2)]
d_t = hx.as_i64(np.concatenate([ins[b % 2][0] for b in range(260)])).to(dev)
d_r = hx.as_i64(np.concatenate([ins[b % 2][1] for b in range(260)])).to(dev)
plan.keyswitch(d_r, d_t, 260)
ctx.sync()
out = hx.to_u64(d_r).reshape(260, -1)
want = [case.expected(orc, t, r) for t, r in ins]
ok = all(np.array_equal(out[b], want[b % 2]) for b in range(260))
d_t, d_r = bench.device_inputs(hx, orc, case, B, dev)
for _ in range(2):
    plan.keyswitch(d_r, d_t, B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    plan.keyswitch(d_r, d_t, B)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"parity={'OK' if ok else 'MISMATCH'} batch={B} L={L} primes just below 2^52 ({mods[0]} ...): {B / dt:,.0f} keyswitch/s ({dt * 1e3:.2f} ms)")
