#!/usr/bin/env python3
"""standalone NTT rates (N = 16384, batch 1024 and 4096, 300 launches) at chosen moduli: usage ntt_q_rate.py [q ...] (default: the
largest prime below 2^52, SURVEY 8d's 2^52 + 393217, the largest below 2^52 * 1.125, a 59-bit prime); HEXL_NTT_INT=1 forces the integer kernels"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import torch
import hexl_fpga_amd as hx
import orc
import bench


def prime_below(v):
    v = ((v - 1) // 32768) * 32768 + 1
    while not orc.orc().orc_is_prime(v):
        v -= 32768
    return v


qs = [int(a) for a in sys.argv[1:]] or [prime_below(1 << 52), 4503599627763713, prime_below((1 << 52) + (1 << 49)), orc.primes(1, 59, 16384)[0]]
dev = torch.device("cuda:0")
ctx = hx.Context(0)
for q in qs:
    for b in (1024, 4096):
        r = bench.time_ntt(hx, ctx, orc, dev, b, 300, q=q)
        print(f"q={q} (2^{q.bit_length() - 1} x {q / 2 ** (q.bit_length() - 1):.4f}) batch {b}: fwd {r['fwd']['ntt_per_s'] / 1e6:.2f} M/s  inv {r['inv']['ntt_per_s'] / 1e6:.2f} M/s")
