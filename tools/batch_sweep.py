#!/usr/bin/env python3
"""batch_sweep.py [L = 7] [batches = 1,2,4,...] -- keyswitch latency / throughput against batch size (N=16384, K = L+1),
device-resident data, the library's default path selection (HEXL_KS_LAT=0 in the environment turns the latency path off)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import torch
import hexl_fpga_amd as hx
import orc
import bench
from ks_util import KsCase

dev = torch.device("cuda:0")
ctx = hx.Context(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
BS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 1024]
case = KsCase(orc, 16384, L, L + 1, seed=1)
plan = hx.KeySwitchPlan(ctx, 16384, L, L + 1, L + 1, 2, case.moduli, case.modswitch)
plan.set_keys(case.keys)
for B in BS:
    d_t, d_r = bench.device_inputs(hx, orc, case, B, dev)
    for _ in range(3):
        plan.keyswitch(d_r, d_t, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20 if B <= 64 else 5
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    e0.record()
    for _ in range(iters):
        plan.keyswitch(d_r, d_t, B)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"batch {B:5d}: {ms * 1e3:9.1f} us per launch  {B / ms * 1e3:10.0f} keyswitch/s")
