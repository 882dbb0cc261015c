import sys
from pathlib import Path
ROOT = Path("/root/repo") if Path("/root/repo/bench.py").exists() else Path(".")
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import torch, hexl_fpga_amd as hx, orc, bench
from ks_util import KsCase
dev = torch.device("cuda:0"); ctx = hx.Context(0)
case = KsCase(orc, 16384, 7, 8, seed=1)
plan = hx.KeySwitchPlan(ctx, 16384, 7, 8, 8, 2, case.moduli, case.modswitch); plan.set_keys(case.keys)
d_t, d_r = bench.device_inputs(hx, orc, case, 1, dev)
for _ in range(20): plan.keyswitch(d_r, d_t, 1)
torch.cuda.synchronize()
