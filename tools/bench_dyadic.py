#!/usr/bin/env python3
"""BASELINE config 3: dyadic_multiply n=8192, 4 RNS moduli, batch 4096 ciphertext pairs resident in HBM.
Prints items/s and achieved algorithmic GB/s (56 B per coefficient-limb = 1,835,008 B per item)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import hexl_fpga_amd as hx
import orc

n, nm = 8192, 4
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ctx = hx.Context(0)
mod1 = np.array(orc.primes(nm, 52, n), dtype=np.uint64)
rng = np.random.default_rng(0)
one = np.concatenate([rng.integers(0, int(m), n, dtype=np.uint64) for _ in range(2) for m in mod1])
a = hx.as_i64(one).to(dev).repeat(batch)
b = hx.as_i64(one[::-1].copy() % np.tile(np.repeat(mod1, n), 2)).to(dev).repeat(batch)
mod = hx.as_i64(np.tile(mod1, batch)).to(dev)
out = torch.empty(batch * 3 * nm * n, dtype=torch.int64, device=dev)
ctx.dyadic_multiply(out, a, b, mod, n, nm)
torch.cuda.synchronize()
ref = orc.dyadic(hx.to_u64(a[: 2 * nm * n]), hx.to_u64(b[: 2 * nm * n]), n, mod1)
assert np.array_equal(hx.to_u64(out[: 3 * nm * n]), ref) and np.array_equal(hx.to_u64(out[-3 * nm * n:]), ref)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 10
e0.record()
for _ in range(iters):
    ctx.dyadic_multiply(out, a, b, mod, n, nm)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
byts = batch * 7 * nm * n * 8
print(f"dyadic n={n} moduli={nm} batch={batch}: {ms:.3f} ms/launch, {batch / ms * 1e3:.0f} items/s, "
      f"{byts / ms / 1e6:.0f} GB/s algorithmic ({byts / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)")
