#!/usr/bin/env python3
"""Twenty single-instance keyswitch launches (N = 16384, L = 6) for `rocprofv3 --kernel-trace`: the five dependent kernels of the
(b, d)-major pipeline run back to back, 18.6 + 17.5 + 16.7 + 14.2 + 5.1 us = 72 us -- one transform's latency each."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'oracle'), str(ROOT / 'tests')]
import torch, hexl_fpga_amd as hx, orc, bench
from ks_util import KsCase
dev=torch.device('cuda:0'); ctx=hx.Context(0)
case=KsCase(orc,16384,6,7,seed=1); plan=hx.KeySwitchPlan(ctx,16384,6,7,7,2,case.moduli,case.modswitch); plan.set_keys(case.keys)
d_t,d_r=bench.device_inputs(hx,orc,case,1,dev)
for _ in range(20): plan.keyswitch(d_r,d_t,1)
torch.cuda.synchronize()
