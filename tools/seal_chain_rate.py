#!/usr/bin/env python3
"""seal_chain_rate.py [batch = 2048] -- bridge-seal's prime chain (52,30,30,40,27,27,27; seal_test.sh:20) on the device: keyswitch/s and
microseconds per lone keyswitch with per-limb arithmetic tiers and with the plan-wide tier (HEXL_KS_PER_LIMB=0); bench.seal_chain_rows"""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import torch
import hexl_fpga_amd as hx
import orc
import bench

ctx = hx.Context(0)
rows = bench.seal_chain_rows(hx, ctx, orc, torch.device("cuda:0"), batch=int(sys.argv[1]) if len(sys.argv) > 1 else 2048)
print(json.dumps(rows, indent=1))
