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


def device_inputs(hx, orc_mod, case, batch, dev, distinct=None):
    """`batch` independent instances generated on the device: every limb uniform in [0, q_i) (torch.randint, seeded),
    t_target[b][L][n] and result[b][2][L][n] as int64 bit patterns of the uint64 words"""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + case.seed)
    n, L = case.n, case.L
    t = torch.empty((batch, L, n), dtype=torch.int64, device=dev)
    r = torch.empty((batch, 2, L, n), dtype=torch.int64, device=dev)
    for i in range(L):
        q = int(case.moduli[i])
        t[:, i].random_(0, q, generator=g)
        r[:, :, i].random_(0, q, generator=g)
    return t.reshape(batch, -1), r.reshape(batch, -1)


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


def cpu_baseline(orc_mod, case, budget_s=10.0):
    """oracle/cpu_baseline.c (Harvey/Shoup port of the reference's CPU algorithms, built -O3 -march=native on THIS host)
    timed on the host cores: one thread, then OpenMP over independent ciphertexts on every core the process may use.
    Bounded sample: ~budget_s seconds per leg. Checked against the line-by-line oracle on one instance first."""
    cb = orc_mod.CpuKeySwitch(case.n, case.L, case.K, case.moduli, case.keys, case.modswitch)
    t1, r1 = case.inputs(orc_mod, 0)
    got = r1.copy()
    cb.keyswitch_batch(got, t1, 1)
    assert np.array_equal(got, case.expected(orc_mod, t1, r1)), "CPU port disagrees with the oracle"
    cores = len(os.sched_getaffinity(0))
    threads = min(cb.lib.cb_max_threads(), cores)
    ts, rs = zip(*[case.inputs(orc_mod, b) for b in range(8)])

    def leg(nthreads, batch):
        t = np.tile(np.concatenate(ts), (batch + 7) // 8)[:batch * case.L * case.n].copy()
        r = np.tile(np.concatenate(rs), (batch + 7) // 8)[:batch * 2 * case.L * case.n].copy()
        cb.keyswitch_batch(r, t, nthreads)                       # warm-up (page faults, thread pool)
        done, t0 = 0, time.perf_counter()
        while True:
            cb.keyswitch_batch(r, t, nthreads)
            done += batch
            el = time.perf_counter() - t0
            if el > budget_s:
                return done / el, done, el

    v1, n1, e1 = leg(1, 8)
    va, na, ea = leg(threads, 2 * threads)
    cb.close()
    return {"value": va, "unit": "keyswitches/s", "cores": threads, "kind": "port", "value_1t": v1,
            "host_cores_visible": cores, "host_cores_total": os.cpu_count(),
            "sample": f"{na} keyswitches N={case.n} L={case.L} K={case.K} in {ea:.1f}s on {threads} OpenMP threads "
                      f"(one ciphertext per thread, affinity mask of {cores} of {os.cpu_count()} cores); 1 thread: {n1} in "
                      f"{e1:.1f}s; oracle/cpu_baseline.c (Harvey/Shoup port of the reference's CPU algorithms, "
                      f"gcc -O3 -march=native -fopenmp; Intel HEXL itself is not in the image)"}


def ctx_cus(ctx):
    import re
    m = re.search(r"(\d+) CUs", ctx.describe())
    return int(m.group(1)) if m else 256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192, help="keyswitches per GPU per step (44 GB of ciphertexts at the default)")
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
    # HEXL_BENCH_ONE_GPU=1 (pre-flight test of the N > 1 path on a one-GPU box, tests/test_gpu_bench_ranks.py): every rank
    # uses GPU 0 and the timing barrier / max-reduce go through gloo (RCCL refuses two ranks on one device)
    one_gpu = os.environ.get("HEXL_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
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

    # in-run check of the measured path: instance 0 of the first full-batch launch against the oracle
    t0_host, r0_host = hx.to_u64(d_t[0]).copy(), hx.to_u64(d_r[0]).copy()
    plan.keyswitch(d_r, d_t, a.batch)
    ctx.sync()
    verified = bool(np.array_equal(hx.to_u64(d_r[0]), case.expected(orc_mod, t0_host, r0_host)))
    assert verified, "keyswitch output differs from the oracle"
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
        tt = torch.tensor([elapsed], device="cpu" if one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    total_ks = a.batch * a.steps * world
    value = total_ks / elapsed
    out = {
        "metric": "keyswitches/sec at N=16384, decomp=7", "value": value, "unit": "keyswitches/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "verified_vs_oracle": verified,
        "config": {"workload": f"keyswitch N={N} decomp_modulus_size={L} key_modulus_size={K} 52-bit primes "
                               f"(GeneratePrimes(K,51,N)), kcc=2, batch {a.batch}/GPU resident in HBM",
                   "parallelism": f"{world} independent shard(s), no collective"},
    }
    if rank == 0:
        alg = ks_alg_bytes(N, L)
        ach = alg * a.batch * a.steps / (dev_ms * 1e-3) / 1e9          # this rank, device-timed
        stage = plan.time_stages(d_r, d_t, min(a.batch, 256), 3)
        us_per_ks = dev_ms * 1e3 / (a.batch * a.steps)
        # Both bounds, recomputable from profiles/ alone (tools/pmc_summary.py writes the two JSON files from the PMC
        # passes of tools/pmc_quick.sh): HBM-side bytes = sum over the pipeline's kernels of 2*FETCH_SIZE + WRITE_SIZE;
        # FP64 issue time = VALU instructions per keyswitch (SQ_INSTS_VALU) x 4 cycles (one wave64 FP64 instruction on
        # a SIMD, MI355X_MICROARCH.md: 78.6 TFLOP/s vector FP64) / (4 SIMDs x CUs) / shader clock under this load
        traffic, traffic_src, alu = None, None, None
        tj = ROOT / "profiles" / "traffic_latest.json"
        if tj.exists():
            t = json.loads(tj.read_text())
            if t.get("L") == L:
                traffic = t["keyswitch_traffic_bytes_per_unit"] * a.batch
                traffic_src = "profiles/traffic_latest.json (PMC passes of tools/pmc_quick.sh, per keyswitch x batch)"
        aj = ROOT / "profiles" / "alu_latest.json"
        if aj.exists():
            t = json.loads(aj.read_text())
            if t.get("L") == L:
                clk = t.get("shader_clock_ghz", 2.0)
                issue_us = t["valu_wave_instructions_per_keyswitch"] * 4.0 / (4 * ctx_cus(ctx)) / (clk * 1e3)
                alu = {"bound": "valu_fp64", "valu_wave_instructions_per_keyswitch": t["valu_wave_instructions_per_keyswitch"],
                       "cycles_per_wave_instruction": 4, "simds": 4 * ctx_cus(ctx), "shader_clock_ghz": clk,
                       "issue_us_per_keyswitch": issue_us, "measured_us_per_keyswitch": us_per_ks,
                       "achieved_frac": issue_us / us_per_ks, "source": "profiles/alu_latest.json (SQ_INSTS_VALU per kernel)"}
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "alg_bytes_per_launch": alg * a.batch,
                           "kernel": "keyswitch pipeline (k_ksx_intt + k_ksx_special + k_ksx_main; chunks of 256 keyswitches)",
                           "alg_bytes_per_keyswitch": alg, "device_ms_per_step": dev_ms / a.steps,
                           # per chunk of 256, from hipEvents on the launch stream (compare avg_us in profiles/*kernel_trace*)
                           "dominant_kernel": {"name": "k_ksx_main (steps 2-3 of the L decomposition limbs, steps 5-7)",
                                               "ms_per_chunk": stage[3], "share_of_pipeline": stage[3] / stage[0],
                                               "chunk": min(a.batch, 256)},
                           # the binding bound: 72 N-point transforms of exact 52-bit arithmetic per 4.6 MB
                           "alu": alu}
        extra = {"stage_ms_at_batch_%d" % min(a.batch, 256): {"total": stage[0], "step_1_inverse_transforms": stage[1],
                                                             "steps_2_4_special_limb": stage[2],
                                                             "steps_2_3_5_7_decomposition_limbs": stage[3]},
                 "device": ctx.describe()}
        if not a.no_extra:
            extra["ntt_N16384_batch1024"] = time_ntt(hx, ctx, orc_mod, dev, 1024, 10)
            extra["dyadic_n8192_m4_batch4096"] = time_dyadic(hx, ctx, orc_mod, dev)
            def other_shape(Lx, Kx, moduli=None, n=N):
                cs = KsCase(orc_mod, n, Lx, Kx, seed=99, moduli=moduli)
                pl = hx.KeySwitchPlan(ctx, n, Lx, Kx, Kx, 2, cs.moduli, cs.modswitch)
                pl.set_keys(cs.keys)
                nbx = min(a.batch, 2048) * (N // n)
                tx, rx = device_inputs(hx, orc_mod, cs, nbx, dev)
                pl.keyswitch(rx, tx, nbx)
                torch.cuda.synchronize()
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
                for _ in range(3):
                    pl.keyswitch(rx, tx, nbx)
                f1.record()
                torch.cuda.synchronize()
                ms = f0.elapsed_time(f1) / 3
                pl.close()
                return {"keyswitches_per_s": nbx / (ms * 1e-3), "batch": nbx,
                        "alg_GBps": ks_alg_bytes(n, Lx) * nbx / (ms * 1e-3) / 1e9}
            # ciphertext multiply + relinearize: DyadicMultiply then KeySwitch as two primitives vs the fused pass
            def mulrelin():
                nbx = min(a.batch, 2048)
                g = torch.Generator(device=dev)
                g.manual_seed(5)
                xa = torch.empty((nbx, 2, L, N), dtype=torch.int64, device=dev)
                xb = torch.empty((nbx, 2, L, N), dtype=torch.int64, device=dev)
                for i in range(L):
                    xa[:, :, i].random_(0, int(case.moduli[i]), generator=g)
                    xb[:, :, i].random_(0, int(case.moduli[i]), generator=g)
                mod = hx.as_i64(np.tile(case.moduli[:L], nbx)).to(dev)
                prod = torch.empty((nbx, 3, L, N), dtype=torch.int64, device=dev)
                out = torch.empty((nbx, 2, L, N), dtype=torch.int64, device=dev)
                tt = torch.empty((nbx, L, N), dtype=torch.int64, device=dev)

                def two_calls():
                    ctx.dyadic_multiply(prod.reshape(-1), xa.reshape(-1), xb.reshape(-1), mod, N, L)
                    out.copy_(prod[:, :2]); tt.copy_(prod[:, 2])          # the caller's re-packing between the primitives
                    plan.keyswitch(out.reshape(-1), tt.reshape(-1), nbx)

                def fused():
                    plan.multiply_relinearize(out.reshape(-1), xa.reshape(-1), xb.reshape(-1), nbx)
                res = {}
                for name, fn in (("dyadic_then_keyswitch", two_calls), ("fused", fused)):
                    fn(); torch.cuda.synchronize()
                    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    f0.record()
                    for _ in range(3):
                        fn()
                    f1.record(); torch.cuda.synchronize()
                    res[name + "_per_s"] = nbx / (f0.elapsed_time(f1) / 3 * 1e-3)
                res["batch"] = nbx
                return res
            extra["multiply_relinearize_16384_L%d" % L] = mulrelin()
            # the reference-representable shape 16384_6_7_7_2 (decomp 6, 7 key moduli), 52-bit primes
            extra["keyswitch_16384_6_7_7_2"] = other_shape(6, 7)
            # the same shape with 48-bit primes (SEAL's default parameter sizes for N=16384): longer lazy-reduction period
            extra["keyswitch_16384_6_7_7_2_48bit_primes"] = other_shape(6, 7, orc_mod.primes(7, 48, N))
            # the smaller ring dimensions the reference's KeySwitch accepts (host/src/keyswitch.cpp:23-25), decomp 3 / 4 key moduli
            for nx in (8192, 4096, 1024):
                extra["keyswitch_%d_3_4_4_2" % nx] = other_shape(3, 4, None, nx)
            # the headline shape on the 64-bit INTEGER kernels (59-bit primes: beyond the reference's < 2^52 envelope)
            extra["keyswitch_16384_L%d_59bit_primes_integer_kernels" % L] = other_shape(L, L + 1, orc_mod.primes(L + 1, 59, N))
        out["extra"] = extra
        if not a.no_cpu and world == 1:                            # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(orc_mod, case)
        print(json.dumps(out))
    plan.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
