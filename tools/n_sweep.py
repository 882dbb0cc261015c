#!/usr/bin/env python3
"""n_sweep.py -- keyswitch throughput against ring dimension (L=3, K=4, 51-bit primes, batch 1024, resident data)."""
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

dev = torch.device("cuda:0")
ctx = hx.Context(0)
L, K = 3, 4
for n in (1024, 2048, 4096, 8192, 16384, 32768):
    case = KsCase(orc, n, L, K, seed=1)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    B = max(256, 1024 * 16384 // n)
    distinct = [case.inputs(orc, b) for b in range(4)]
    ts = np.concatenate([distinct[b % 4][0] for b in range(B)])
    rs = np.concatenate([distinct[b % 4][1] for b in range(B)])
    d_t, d_r = hx.as_i64(ts).to(dev), hx.as_i64(rs).to(dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        plan.keyswitch(d_r, d_t, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        plan.keyswitch(d_r, d_t, B)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    # transforms per keyswitch: L inverse + L*L forward + 2 inverse + 2L forward
    tr = L + L * L + 2 + 2 * L
    print(f"n={n:6d} batch {B:6d}: {B / ms * 1e3:10.0f} keyswitch/s  {B * tr * n / ms / 1e6:8.1f} G coefficient-transforms/s")
    plan.close()
