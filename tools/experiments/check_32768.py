import sys
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np, torch
import hexl_fpga_amd as hx, orc
from ks_util import KsCase
dev = torch.device("cuda:0"); ctx = hx.Context(0)
n = 32768
for bits in (51, 30, 55):
    q = orc.primes(1, bits, n)[0]
    tb = orc.HexlTables(n, q)
    xs = np.stack([orc.splitmix(n, 10 + b, q) for b in range(5)])
    d = hx.as_i64(xs).to(dev)
    tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
    ctx.ntt_fwd(d, tabs[0], tabs[1], q, n); ctx.sync()
    ok_f = np.array_equal(hx.to_u64(d), orc.ntt_fwd(xs, tb))
    ctx.ntt_inv(d, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, n); ctx.sync()
    ok_i = np.array_equal(hx.to_u64(d), xs)
    print(f"n={n} {bits}-bit prime: forward {'OK' if ok_f else 'MISMATCH'}, round trip {'OK' if ok_i else 'MISMATCH'}")
for (L, K) in ((2, 3), (6, 7)):
    case = KsCase(orc, n, L, K, seed=5)
    plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch); plan.set_keys(case.keys)
    ins = [case.inputs(orc, b) for b in range(3)]
    d_t = hx.as_i64(np.concatenate([t for t, _ in ins])).to(dev); d_r = hx.as_i64(np.concatenate([r for _, r in ins])).to(dev)
    plan.keyswitch(d_r, d_t, 3); ctx.sync()
    out = hx.to_u64(d_r).reshape(3, -1)
    print(f"keyswitch n={n} L={L} K={K}:", all(np.array_equal(out[b], case.expected(orc, *ins[b])) for b in range(3)))
    plan.close()
