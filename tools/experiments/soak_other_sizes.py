import sys
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np, torch, hexl_fpga_amd as hx, orc, bench
from ks_util import KsCase
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0"); ctx = hx.Context(0); bad = 0
for n, L, K, B in ((32768, 3, 4, 160), (8192, 6, 7, 700), (4096, 3, 4, 1500), (1024, 2, 3, 4000)):
    case = KsCase(orc, n, L, K, seed=L + n)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch); plan.set_keys(case.keys)
    ins = [case.inputs(orc, b) for b in range(4)]
    d_t = hx.as_i64(np.concatenate([ins[b % 4][0] for b in range(B)])).to(dev)
    d_r0 = hx.as_i64(np.concatenate([ins[b % 4][1] for b in range(B)])).to(dev)
    ref = None
    for it in range(iters):
        d_r = d_r0.clone(); plan.keyswitch(d_r, d_t, B); ctx.sync()
        if ref is None:
            ref = d_r.clone()
            out = hx.to_u64(d_r).reshape(B, -1)
            assert all(np.array_equal(out[b], case.expected(orc, *ins[b % 4])) for b in (0, 1, 2, 3, B - 1))
        elif not torch.equal(ref, d_r):
            bad += 1; print("MISMATCH", n, it)
    plan.close(); print(f"keyswitch n={n} L={L} batch={B}: {iters} identical runs, oracle-checked")
n = 32768; q = orc.primes(1, 51, n)[0]; tb = orc.HexlTables(n, q)
tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
x0 = hx.as_i64(np.stack([orc.splitmix(n, 77 + b, q) for b in range(512)])).to(dev).contiguous(); ref = None
for it in range(iters):
    x = x0.clone(); ctx.ntt_fwd(x, tabs[0], tabs[1], q, n); ctx.sync()
    if ref is None: ref = x.clone()
    elif not torch.equal(ref, x): bad += 1; print("MISMATCH fwd", it)
    ctx.ntt_inv(x, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, n); ctx.sync()
    if not torch.equal(x, x0): bad += 1; print("MISMATCH round trip", it)
print("NTT n=32768 batch 512:", iters, "runs; mismatches:", bad)
sys.exit(1 if bad else 0)
