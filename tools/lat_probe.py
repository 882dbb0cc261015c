#!/usr/bin/env python3
"""lat_probe.py [L = 6] [seconds = 2] -- a lone keyswitch (N = 16384) launched back to back for a while, with the board power and
shader clock sampled (bench.PowerSampler): device time per keyswitch, and the clock the mostly idle chip runs the latency path at."""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import torch
import hexl_fpga_amd as hx
import orc
import bench
from ks_util import KsCase

L = int(sys.argv[1]) if len(sys.argv) > 1 else 6
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
dev = torch.device("cuda:0")
ctx = hx.Context(0)
case = KsCase(orc, 16384, L, L + 1, seed=1)
plan = hx.KeySwitchPlan(ctx, 16384, L, L + 1, L + 1, 2, case.moduli, case.modswitch)
plan.set_keys(case.keys)
d_t, d_r = bench.device_inputs(hx, orc, case, 1, dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(50):
    plan.keyswitch(d_r, d_t, 1)
torch.cuda.synchronize()
ps = bench.PowerSampler(0)
ps.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n, t0 = 0, time.perf_counter()
e0.record()
while time.perf_counter() - t0 < secs:
    for _ in range(200):
        plan.keyswitch(d_r, d_t, 1)
    n += 200
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
p = ps.stop()
print(f"L={L}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per lone keyswitch over {n} launches; board {p and p['board_power_w_mean']:.0f} W, "
      f"shader clock {p and p['sclk_mhz_mean']:.0f} MHz")
