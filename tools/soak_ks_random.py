#!/usr/bin/env python3
"""Randomised parity soak of the keyswitch against the oracle: random primes = 1 mod 2n of random sizes in [2^17, 2^52) (so every
FP64 tier and every mix of tiers across the limbs of one plan comes up), random ring dimension, decomposition size and batch --
batches on both sides of the slot-major threshold, so the slot-major, (b, d)-major and latency kernels all run --, uniform and
worst-case (ks_util.extreme_words) keys and inputs; one plan in twelve has 53 ... 59-bit primes (integer kernels); a quarter of the FP64 cases
go through hexl_multiply_relinearize against the composition of the two oracle calls. Every instance of a batch is one of three distinct
ones; ALL are compared, on the device, over several launches of the same case (a race shows up in one instance of one launch).
usage: soak_ks_random.py [seconds = 240] [seed] [launches per case = 3]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc
from ks_util import KsCase

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260930
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
ctx = hx.Context(0)
rng = np.random.default_rng(seed)
EDGES = [1 << 49, 1 << 50, (1 << 51) + (1 << 44), 1 << 52]          # tier tops (f64_arith.hpp): period 12 / 6 / 3 / strict


def random_prime(n, used):
    while True:
        kind = rng.integers(0, 3)
        if kind == 0:                                               # just below a tier top
            v = int(EDGES[rng.integers(0, 4)]) - int(rng.integers(1, 1 << 30))
        elif kind == 1:                                             # just above a tier top (not above 2^52)
            v = int(EDGES[rng.integers(0, 3)]) + int(rng.integers(1, 1 << 30))
        else:                                                       # any size from 27 bits up
            b = int(rng.integers(18, 53))                           # (hexl_ks_plan_create takes moduli from 2^16 up)
            v = int(rng.integers(1 << (b - 1), 1 << b))
        v = v // (2 * n) * (2 * n) + 1
        while v > (1 << 17) and (v in used or not orc.orc().orc_is_prime(v)):
            v -= 2 * n
        if v > (1 << 17) and v < (1 << 52):
            return v


t0 = time.time()
cases = fails = 0
seen_tiers = set()
prev = None                                                       # the previous case, kept alive: its plan is launched again between this case's launches
while time.time() - t0 < budget:
    n = int(rng.choice([1024, 2048, 4096, 8192, 16384, 16384, 16384, 32768]))
    L = int(rng.integers(1, 8))
    K = L + 1 + int(rng.integers(0, 2)) * int(rng.integers(0, 3))
    moduli = []
    integer = rng.integers(0, 12) == 0 and n <= 16384             # (the integer kernels stop at N = 16384)
    if integer:
        moduli = orc.primes(K, int(rng.integers(53, 60)), n)
    else:
        for _ in range(K):
            moduli.append(random_prime(n, moduli))
    per_cu = max(1, 16384 // n)
    nb = int(rng.choice([1, 2, 3, 5, 17, 40 * per_cu, 70 * per_cu, 300 * per_cu // max(1, L // 2)]))
    extreme = bool(rng.integers(0, 2))
    case = KsCase(orc, n, L, K, seed=int(rng.integers(1, 1 << 20)), moduli=moduli, extreme_keys=extreme)
    try:
        plan = hx.KeySwitchPlan(ctx, n, L, K, K, 2, case.moduli, case.modswitch)
    except hx.HexlError:                                          # (HEXL_KS_INT=1 at N = 32768: no integer kernels there)
        continue
    plan.set_keys(case.keys)
    seen_tiers.update(plan.tiers()[0] if hasattr(plan, "tiers") else [])
    ins = [case.extreme_inputs(orc, b) if extreme else case.inputs(orc, b) for b in range(3)]
    idx = torch.arange(nb, device=dev) % 3
    fused = (-1 not in plan.tiers()[0]) and rng.integers(0, 4) == 0     # (the integer kernels -- moduli >= 2^52 or HEXL_KS_INT=1 -- have no fused pass)
    if fused:                                                      # out = (a0 b0, a0 b1 + a1 b0) + KeySwitch(a1 b1): operands = the result-shaped words
        A = [r for _, r in ins]
        B = [np.roll(r, 1) for _, r in ins]
        for i in range(L):                                           # (rolled across a limb boundary: bring the word back below its modulus)
            for Bv in B:
                Bv[i * n] %= np.uint64(moduli[i]); Bv[(L + i) * n] %= np.uint64(moduli[i])
        want = []
        for a_, b_ in zip(A, B):
            prod = orc.dyadic(a_, b_, n, case.moduli[:L], exact=True)
            o = prod[:2 * L * n].copy()
            orc.keyswitch(o, prod[2 * L * n:].copy(), n, L, K, L + 1, case.moduli, case.keys, case.modswitch)
            want.append(o)
        d_a = torch.from_numpy(np.stack(A).view(np.int64)).to(dev)[idx].contiguous()
        d_b = torch.from_numpy(np.stack(B).view(np.int64)).to(dev)[idx].contiguous()
    else:
        want = [case.expected(orc, t, r) for t, r in ins]
        d_t = torch.from_numpy(np.stack([t for t, _ in ins]).view(np.int64)).to(dev)[idx].contiguous()
        d_r0 = torch.from_numpy(np.stack([r for _, r in ins]).view(np.int64)).to(dev)[idx].contiguous()
    d_want = torch.from_numpy(np.stack(want).view(np.int64)).to(dev)[idx]
    ok = True
    for _ in range(launches):
        if fused:
            d_r = torch.full((nb, 2 * L * n), -1, dtype=torch.int64, device=dev)
            plan.multiply_relinearize(d_r, d_a, d_b, nb)
        else:
            d_r = d_r0.clone()
            plan.keyswitch(d_r, d_t, nb)
        if prev is not None:                                       # two plans on one context, launched back to back (scratch growth, plan switches)
            p_r = prev["r0"].clone()
            prev["plan"].keyswitch(p_r, prev["t"], prev["nb"])
        ctx.sync()
        wrong = int((d_r.view(nb, -1) != d_want.view(nb, -1)).any(dim=1).sum())
        if prev is not None:
            pw = int((p_r.view(prev["nb"], -1) != prev["want"].view(prev["nb"], -1)).any(dim=1).sum())
            if pw:
                ok = False
                print(f"  {pw} of {prev['nb']} instances of the PREVIOUS case's plan wrong when interleaved ({prev['desc']})", flush=True)
        if wrong:
            ok = False
            print(f"  {wrong} of {nb} instances wrong in one launch", flush=True)
    cases += 1
    if not ok:
        fails += 1
        print(f"MISMATCH n={n} L={L} K={K} nb={nb} extreme={extreme} fused={fused} moduli={moduli}", flush=True)
    if prev is not None:
        prev["plan"].close()
        prev = None
    if not fused and rng.integers(0, 2):
        prev = {"plan": plan, "r0": d_r0, "t": d_t, "nb": nb, "want": d_want, "desc": f"n={n} L={L} K={K} nb={nb} moduli={moduli}"}
    else:
        plan.close()
print(f"{cases} random keyswitch cases in {time.time() - t0:.0f} s (seed {seed}), tiers seen {sorted(seen_tiers)}, mismatches: {fails}")
sys.exit(1 if fails else 0)
