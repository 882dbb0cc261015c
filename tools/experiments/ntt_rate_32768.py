import sys
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np, torch, hexl_fpga_amd as hx, orc
dev = torch.device("cuda:0"); ctx = hx.Context(0)
for n, batch in ((16384, 1024), (32768, 512)):
    q = orc.primes(1, 51, n)[0]; tb = orc.HexlTables(n, q)
    x = hx.as_i64(np.stack([orc.splitmix(n, 1000 + b, q) for b in range(8)])).to(dev).repeat(batch // 8, 1).contiguous()
    tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
    for name in ("fwd", "inv"):
        run = (lambda: ctx.ntt_fwd(x, tabs[0], tabs[1], q, n)) if name == "fwd" else (lambda: ctx.ntt_inv(x, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, n))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"n={n} {name}: {batch / ms * 1e3 / 1e6:.2f} M NTT/s  {batch * 2 * n * 8 / ms / 1e6:.0f} GB/s")
