#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 (kernel trace or PMC passes): 2 keyswitch launches at batch B (one scratch
chunk) + 2 fwd / 2 inv NTT launches at batch 1024. usage: profile_ks.py [batch] [decomp]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda:0")
ctx = hx.Context(0)
case = KsCase(orc, 16384, L, L + 1, seed=1)
plan = hx.KeySwitchPlan(ctx, 16384, L, L + 1, L + 1, 2, case.moduli, case.modswitch)
plan.set_keys(case.keys)
d_t, d_r = bench.device_inputs(hx, orc, case, B, dev)
for _ in range(2):
    plan.keyswitch(d_r, d_t, B)
torch.cuda.synchronize()
print("ntt", bench.time_ntt(hx, ctx, orc, dev, 1024, 2))
