#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hexl-fpga hot path on MI355X.

Metric (BASELINE.json): keyswitches/sec at N=16384, decomp_modulus_size=7 (key_modulus_size=8,
52-bit primes), data resident in HBM; one "step" = one hexl_keyswitch() pass over the rank's batch of
synthetic ciphertexts. Ranks (one per GPU) each own an independent shard of the batch -- no collective on
the data path -- so `scaling` is weak and `value` = all ranks' keyswitches / max-over-ranks time.

    python bench.py [--gpus N --steps K --warmup W --batch B]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 with the contract fields plus `roofline`, `cpu_baseline` and `extra`
(fwd/inv NTT rates, per-stage kernel times, the reference-representable L=6/K=7 shape).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "oracle"))   # checker + cpu_baseline leg only

N = 16384
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak ~6290


def ks_alg_bytes(n, L):
    """compulsory HBM bytes per keyswitch: read t_target[L][n], read+write result[2][L][n]"""
    return (L + 2 * 2 * L) * n * 8


def device_inputs(hx, orc_mod, case, batch, dev, distinct=8):
    """batch instances built from `distinct` independent splitmix instances (keeps host prep cheap; every
    instance is still full-entropy data mod its limb)"""
    import torch
    ts, rs = zip(*[case.inputs(orc_mod, b) for b in range(min(distinct, batch))])
    t = hx.as_i64(np.stack(ts)).to(dev)
    r = hx.as_i64(np.stack(rs)).to(dev)
    reps = (batch + t.shape[0] - 1) // t.shape[0]
    return t.repeat(reps, 1)[:batch].contiguous(), r.repeat(reps, 1)[:batch].contiguous()


def time_ntt(hx, ctx, orc_mod, dev, batch, iters):
    import torch
    q = orc_mod.primes(1, 51, N)[0]
    tb = orc_mod.HexlTables(N, q)
    x = hx.as_i64(np.stack([orc_mod.splitmix(N, 1000 + b, q) for b in range(8)])).to(dev)
    x = x.repeat((batch + 7) // 8, 1)[:batch].contiguous()
    tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
    out = {}
    for name in ("fwd", "inv"):
        def run():
            if name == "fwd":
                ctx.ntt_fwd(x, tabs[0], tabs[1], q, N)
            else:
                ctx.ntt_inv(x, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, N)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out[name] = {"ms_per_launch": ms, "ntt_per_s": batch / (ms * 1e-3),
                     "alg_GBps": batch * 2 * N * 8 / (ms * 1e-3) / 1e9}
    return out


def time_dyadic(hx, ctx, orc_mod, dev, batch=4096, n=8192, nm=4, iters=5):
    """BASELINE config 3: dyadic_multiply n=8192, 4 RNS moduli, batch 4096 ciphertext pairs (56 B per coefficient-limb)"""
    import torch
    mod1 = np.array(orc_mod.primes(nm, 52, n), dtype=np.uint64)
    rng = np.random.default_rng(0)
    one = np.concatenate([rng.integers(0, int(m), n, dtype=np.uint64) for _ in range(2) for m in mod1])
    a = hx.as_i64(one).to(dev).repeat(batch)
    b = hx.as_i64(one[::-1].copy() % np.tile(np.repeat(mod1, n), 2)).to(dev).repeat(batch)
    mod = hx.as_i64(np.tile(mod1, batch)).to(dev)
    out = torch.empty(batch * 3 * nm * n, dtype=torch.int64, device=dev)
    ctx.dyadic_multiply(out, a, b, mod, n, nm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ctx.dyadic_multiply(out, a, b, mod, n, nm)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return {"ms_per_launch": ms, "items_per_s": batch / (ms * 1e-3), "alg_GBps": batch * 7 * nm * n * 8 / (ms * 1e-3) / 1e9}


def cpu_baseline(orc_mod, case, budget_s=12.0):
    """the oracle (C restatement of the reference algorithm, single thread) timed on this host"""
    t, r = case.inputs(orc_mod, 0)
    done, t0 = 0, time.perf_counter()
    while True:
        case.expected(orc_mod, t, r)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s or done >= 1024:
            break
    return {"value": done / el, "unit": "keyswitches/s", "cores": 1, "kind": "port",
            "sample": f"{done} keyswitch(es) N={case.n} L={case.L} K={case.K}, oracle/hexl_oracle.c -O3, 1 thread of "
                      f"{os.cpu_count()} host cores, {el:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="keyswitches per GPU per step")
    ap.add_argument("--decomp", type=int, default=7, help="decomp_modulus_size L (key_modulus_size = L+1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hexl_fpga_amd as hx
    import orc as orc_mod
    from ks_util import KsCase

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = hx.Context(local)
    L, K = a.decomp, a.decomp + 1
    case = KsCase(orc_mod, N, L, K, seed=1 + rank)          # every rank: its own shard of ciphertexts
    plan = hx.KeySwitchPlan(ctx, N, L, K, L + 1, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    d_t, d_r = device_inputs(hx, orc_mod, case, a.batch, dev)

    for _ in range(a.warmup):
        plan.keyswitch(d_r, d_t, a.batch)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        plan.keyswitch(d_r, d_t, a.batch)
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)                              # HIP events on the launch stream
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    total_ks = a.batch * a.steps * world
    value = total_ks / elapsed
    out = {
        "metric": "keyswitches/sec at N=16384, decomp=7", "value": value, "unit": "keyswitches/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"keyswitch N={N} decomp_modulus_size={L} key_modulus_size={K} 52-bit primes "
                               f"(GeneratePrimes(K,51,N)), kcc=2, batch {a.batch}/GPU resident in HBM",
                   "parallelism": f"{world} independent shard(s), no collective"},
    }
    if rank == 0:
        alg = ks_alg_bytes(N, L)
        ach = alg * a.batch * a.steps / (dev_ms * 1e-3) / 1e9          # this rank, device-timed
        stage = plan.time_stages(d_r, d_t, min(a.batch, 256), 3)
        # HBM-side bytes per launch from the committed PMC passes (tools/pmc_summary.py: 2*FETCH_SIZE + WRITE_SIZE
        # summed over the pipeline's kernels, separate --pmc runs); scaled from the profiled batch to this batch
        traffic = None
        tj = ROOT / "profiles" / "traffic_latest.json"
        if tj.exists():
            t = json.loads(tj.read_text())
            if t.get("L") == L:
                traffic = t["keyswitch_traffic_bytes_per_unit"] * a.batch
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "alg_bytes_per_launch": alg * a.batch,
                           "kernel": "keyswitch pipeline (k_ksf_up + k_ksf_mac + k_ksf_intt_sp + k_ksf_moddown)",
                           "alg_bytes_per_keyswitch": alg, "device_ms_per_step": dev_ms / a.steps,
                           # the pipeline runs in chunks of 256 keyswitches; per chunk, from hipEvents on the launch
                           # stream (compare avg_us of the same kernels in profiles/*kernel_trace*)
                           "dominant_kernel": {"name": "k_ksf_up (steps 1-2)", "ms_per_chunk": stage[1],
                                               "share_of_pipeline": stage[1] / stage[0],
                                               "chunk": min(a.batch, 256)}}
        extra = {"stage_ms_at_batch_%d" % min(a.batch, 256): {"total": stage[0], "steps_1_2_inverse_and_modup": stage[1],
                                                             "steps_3_4_mac_and_special_inverse": stage[2],
                                                             "steps_5_7_moddown": stage[3]},
                 "device": ctx.describe()}
        if not a.no_extra:
            extra["ntt_N16384_batch1024"] = time_ntt(hx, ctx, orc_mod, dev, 1024, 10)
            extra["dyadic_n8192_m4_batch4096"] = time_dyadic(hx, ctx, orc_mod, dev)
            def other_shape(Lx, Kx, moduli=None):
                cs = KsCase(orc_mod, N, Lx, Kx, seed=99, moduli=moduli)
                pl = hx.KeySwitchPlan(ctx, N, Lx, Kx, Kx, 2, cs.moduli, cs.modswitch)
                pl.set_keys(cs.keys)
                tx, rx = device_inputs(hx, orc_mod, cs, a.batch, dev)
                pl.keyswitch(rx, tx, a.batch)
                torch.cuda.synchronize()
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
                for _ in range(3):
                    pl.keyswitch(rx, tx, a.batch)
                f1.record()
                torch.cuda.synchronize()
                ms = f0.elapsed_time(f1) / 3
                pl.close()
                return {"keyswitches_per_s": a.batch / (ms * 1e-3),
                        "alg_GBps": ks_alg_bytes(N, Lx) * a.batch / (ms * 1e-3) / 1e9}
            # the reference-representable shape 16384_6_7_7_2 (decomp 6, 7 key moduli), 52-bit primes
            extra["keyswitch_16384_6_7_7_2"] = other_shape(6, 7)
            # the same shape with 48-bit primes (SEAL's default parameter sizes for N=16384): longer lazy-reduction period
            extra["keyswitch_16384_6_7_7_2_48bit_primes"] = other_shape(6, 7, orc_mod.primes(7, 48, N))
        out["extra"] = extra
        if not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(orc_mod, case)
        print(json.dumps(out))
    plan.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
