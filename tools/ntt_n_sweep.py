#!/usr/bin/env python3
"""standalone _NTT / _INTT throughput against ring dimension (51-bit prime, 128 MiB of polynomials per launch)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc

dev = torch.device("cuda:0")
ctx = hx.Context(0)
for n in (1024, 2048, 4096, 8192, 16384, 32768):
    q = orc.primes(1, 51, n)[0]
    tb = orc.HexlTables(n, q)
    batch = (1 << 24) // n
    x = hx.as_i64(np.stack([orc.splitmix(n, 1000 + b, q) for b in range(8)])).to(dev).repeat(batch // 8, 1).contiguous()
    tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
    res = []
    for name in ("fwd", "inv"):
        def run():
            if name == "fwd":
                ctx.ntt_fwd(x, tabs[0], tabs[1], q, n)
            else:
                ctx.ntt_inv(x, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, n)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(f"{name} {batch / ms * 1e3 / 1e6:7.2f} M/s {batch * 2 * n * 8 / ms / 1e6:6.0f} GB/s")
    print(f"n={n:6d} batch {batch:6d}: " + "   ".join(res))
