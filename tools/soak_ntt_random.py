#!/usr/bin/env python3
"""Randomised parity soak of the standalone _NTT / _INTT against the oracle: random primes = 1 mod 2n of 20 ... 62 bits (every FP64 tier, the
strict extension above 2^52, the integer kernels), every ring dimension, single polynomials and persistent-kernel batches, canonical words
with the extremes mixed in and (in a third of the cases) a few out-of-range words. EVERY polynomial of every launch is compared (on the
device), and every case is launched several times: a race shows up in one polynomial of one launch.
usage: soak_ntt_random.py [seconds = 200] [seed] [launches per case = 6]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc
from ks_util import extreme_words

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260930
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
ctx = hx.Context(0)
rng = np.random.default_rng(seed)
EDGES = [1 << 49, 1 << 50, (1 << 51) + (1 << 44), 1 << 52, (1 << 52) + (1 << 49)]
t0 = time.time()
cases = fails = polys = 0
while time.time() - t0 < budget:
    n = int(rng.choice([1024, 2048, 2048, 4096, 4096, 8192, 8192, 16384, 16384, 32768]))
    if rng.integers(0, 2):
        v = int(EDGES[rng.integers(0, 5)]) + int(rng.integers(-(1 << 30), 1 << 30))
    else:
        b = int(rng.integers(20, 63))
        v = int(rng.integers(1 << (b - 1), 1 << b))
    v = max(v, 4 * n) // (2 * n) * (2 * n) + 1
    while not orc.orc().orc_is_prime(v):
        v += 2 * n
    q = v
    if q >= (1 << 62):
        continue
    t = orc.HexlTables(n, q)
    if rng.integers(0, 6) == 0:                                    # tables that are NOT Shoup tables (benchmark/bench_fwd_ntt.cpp:36-42): the integer
        for a in (t.roots, t.precon, t.inv_roots, t.inv_precon):   # butterflies replay them op for op; also exercises the hint / violation-counter logic,
            a[:] = orc.splitmix(n, int(rng.integers(1, 1 << 30)), q)   # the allocator hands the next case the same device addresses with other contents
    nuniq = 7
    base = np.stack([orc.splitmix(n, int(rng.integers(1, 1 << 30)), q) for _ in range(nuniq)])
    base[0, :4] = np.array([q - 1, 0, q // 2, q // 2 + 1], dtype=np.uint64)
    base[1] = extreme_words(n, q, int(rng.integers(0, 9)))
    if rng.integers(0, 3) == 0:                                    # out-of-range words (still inside the 64-bit contract: the oracle replays them)
        base[2, int(rng.integers(0, n))] = np.uint64(min(2 * q - 1, (1 << 64) - 1))
        base[3, int(rng.integers(0, n))] = np.uint64(min(4 * q - 1, (1 << 64) - 1))
    per_cu = max(1, 16384 // n)
    batch = int(rng.choice([1, 3, 40, 260 * per_cu, 700 * per_cu, 1500 * per_cu]))
    x = torch.from_numpy(base.view(np.int64)).to(dev)[torch.arange(batch, device=dev) % nuniq].contiguous()
    tabs = [hx.as_i64(a).to(dev) for a in (t.roots, t.precon, t.inv_roots, t.inv_precon)]
    for fwd in (True, False):
        want = torch.from_numpy((orc.ntt_fwd if fwd else orc.ntt_inv)(base, t).view(np.int64)).to(dev)[torch.arange(batch, device=dev) % nuniq]
        for _ in range(launches):
            d = x.clone()
            if fwd:
                ctx.ntt_fwd(d, tabs[0], tabs[1], q, n)
            else:
                ctx.ntt_inv(d, tabs[2], tabs[3], q, t.inv_n, t.inv_n_w, n)
            ctx.sync()
            wrong = int((d.view(batch, n) != want.view(batch, n)).any(dim=1).sum())
            polys += batch
            if wrong:
                fails += 1
                print(f"MISMATCH n={n} q={q} ({q.bit_length()} bits) batch={batch} fwd={fwd}: {wrong} polynomials", flush=True)
        cases += 1
print(f"soak_ntt_random: {cases} cases, {polys} polynomials in {time.time() - t0:.0f} s (seed {seed}), mismatching launches: {fails}")
sys.exit(1 if fails else 0)
