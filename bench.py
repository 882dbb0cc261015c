#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hexl-fpga hot path on MI355X.

Metric (BASELINE.json): keyswitches/sec at N=16384, decomp_modulus_size=7 (key_modulus_size=8,
52-bit primes), data resident in HBM; one "step" = one hexl_keyswitch() pass over the rank's shard of the
step's batch of synthetic ciphertexts. Ranks (one per GPU) each own an independent contiguous shard
(hexl_fpga_amd.sharding.shard_range) -- no collective on the data path; `value` = all ranks'
keyswitches / max-over-ranks wall time.

    python bench.py [--gpus N --steps K --warmup W] [--total-batch T | --batch B]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Default = BASELINE config 5 as stated: ONE batch of 8192 ciphertexts per step split over the ranks (8192 on one GPU,
1024 per GPU on eight) -- `scaling: "strong"`. `--batch B` instead gives every rank B ciphertexts per step
(`scaling: "weak"`). `--barrier-per-step` synchronises all ranks after every step (pre-flight of the regime where a
step is 5 ms: tests/test_gpu_bench_ranks.py).

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (its `traffic` and `alu` inputs are PMC passes
of tools/pmc_workload run INSIDE this benchmark at N = 1; --no-pmc falls back to profiles/*_latest.json; `roofline.power` =
board power and shader clock sampled from hwmon over the timed region), `cpu_baseline`
and `extra` (fwd/inv NTT rates over ALL ranks -- BASELINE's second metric --, per-stage kernel times, the
reference-representable L=6/K=7 shape).
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


# the three standalone-NTT workloads reported at N = 16384 (all fwd + inv):
NTT_Q_FAST = None                   # primes(1, 51, N)[0] = 2251799814045697: q in (2^51, 2^52), genuine Shoup tables -> exact FP64 fast path
NTT_Q_SURVEY = 4503599627763713     # SURVEY 8d cfg1/cfg2's prime, 2^52 + 393217: above the lazy FP64 range -> STRICT FP64 kernels (moduli < 2^52 * 1.125)
NTT_BITS_INTEGER = 59               # a modulus only the integer Harvey kernels take (>= 2^52 * 1.125)
NTT_Q_REFBENCH = 136314881          # benchmark/bench_fwd_ntt.cpp:28-42, bench_inv_ntt.cpp: RANDOM roots / precons / inv_n -> integer butterflies


def time_ntt(hx, ctx, orc_mod, dev, batch, iters, barrier=None, max_over_ranks=None, world=1, q=None, random_tables=False, n=None):
    """BASELINE config 2 shape (fwd / inv NTT, N = 16384, one prime, `batch` polynomials per launch) on EVERY rank:
    per-rank device time from HIP events, and -- BASELINE's second metric -- the whole-job rate = world x batch x iters /
    the slowest rank's wall time between two barriers. `random_tables`: the reference benchmark's own workload (uniform random
    words below q as roots, precons, inv_n, inv_n_w: not Shoup tables, so every polynomial takes the integer butterflies)."""
    import torch
    N = n or globals()["N"]
    q = q or orc_mod.primes(1, 51, N)[0]
    if random_tables:
        rng = np.random.default_rng(7)
        class tb: pass
        tb.roots, tb.precon, tb.inv_roots, tb.inv_precon = (rng.integers(0, q, N, dtype=np.uint64) for _ in range(4))
        tb.inv_n, tb.inv_n_w = int(rng.integers(0, q)), int(rng.integers(0, q))
    else:
        tb = orc_mod.HexlTables(N, q)
    x = hx.as_i64(np.stack([orc_mod.splitmix(N, 1000 + b, q) for b in range(8)])).to(dev)
    x = x.repeat((batch + 7) // 8, 1)[:batch].contiguous()
    tabs = [hx.as_i64(a).to(dev) for a in (tb.roots, tb.precon, tb.inv_roots, tb.inv_precon)]
    out = {"q": int(q), "tables": "random words below q (benchmark/bench_fwd_ntt.cpp:36-42)" if random_tables else "Shoup tables of q"}
    for name in ("fwd", "inv"):
        def run():
            if name == "fwd":
                ctx.ntt_fwd(x, tabs[0], tabs[1], q, N)
            else:
                ctx.ntt_inv(x, tabs[2], tabs[3], q, tb.inv_n, tb.inv_n_w, N)
        run()
        torch.cuda.synchronize()
        run()                                                     # (random tables: the second call is the first on the hinted route)
        torch.cuda.synchronize()
        if barrier:
            barrier()
        # device time per launch = the MEDIAN of four event-timed blocks (a descheduled launch thread -- the pods run under a CPU
        # quota -- otherwise lands in the one number); the whole-job rate below is wall time over ALL launches, stalls included
        nblk = 4
        per = max(1, iters // nblk)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(nblk + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for b in range(nblk):
            for _ in range(per):
                run()
            ev[b + 1].record()
        if barrier:
            barrier()
        else:
            torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if max_over_ranks:
            wall = max_over_ranks(wall)
        iters_done = nblk * per
        blocks = sorted(ev[b].elapsed_time(ev[b + 1]) / per for b in range(nblk))
        ms = 0.5 * (blocks[nblk // 2 - 1] + blocks[nblk // 2])
        out[name] = {"ms_per_launch": ms, "ms_per_launch_blocks": blocks, "ntt_per_s": batch / (ms * 1e-3),
                     "alg_GBps": batch * 2 * N * 8 / (ms * 1e-3) / 1e9,
                     "ntt_per_s_all_ranks": world * batch * iters_done / wall, "n_gpus": world}
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


def cpu_quota_cores():
    """the container's CFS CPU quota in cores (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None: threads beyond it are throttled,
    which is what the timed legs and the STREAM probe of `cpu_baseline` show on the pods of this pool"""
    try:
        parts = open("/sys/fs/cgroup/cpu.max").read().split()
        if parts and parts[0] != "max":
            return float(parts[0]) / float(parts[1])
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / p
    except Exception:
        pass
    return None


def cpu_baseline(orc_mod, case, budget_s=24.0):
    """oracle/cpu_baseline.c (Harvey/Shoup port of the reference's CPU algorithms with the AVX-512 kernels HEXL's rule selects, built
    -O3 -march=native on THIS host) timed on the host cores with its NUMA-aware leg (cb_keyswitch_timed, round 4): every OpenMP
    thread pinned to its own core, with a private first-touched copy of its ciphertexts and scratch, and the keys / key Shoup
    factors / twiddle tables (33 MB) replicated once per NUMA node -- round 3's single copy, first-touched by one thread, fed all
    128 threads out of one node's memory (3.9 % parallel efficiency). Legs: 1, 16, 64 threads, one per physical core, one per
    hardware thread; `value` = the best. Bounded sample: ~budget_s seconds in all. Each leg's first keyswitch is checked against
    the line-by-line oracle."""
    cb = orc_mod.CpuKeySwitch(case.n, case.L, case.K, case.moduli, case.keys, case.modswitch)
    t1, r1 = case.inputs(orc_mod, 0)
    want = case.expected(orc_mod, t1, r1)
    got = r1.copy()
    cb.keyswitch_batch(got, t1, 1)
    assert np.array_equal(got, want), "CPU port disagrees with the oracle"
    isa = cb.isa()
    cores = len(os.sched_getaffinity(0))
    # omp_get_max_threads() inside this process is the OpenMP runtime torch has already configured: one thread per PHYSICAL core
    # (128 on the 2 x 64-core, 256-hardware-thread boxes of this pool)
    physical = min(cb.lib.cb_max_threads(), cores)
    ts, rs = zip(*[case.inputs(orc_mod, b) for b in range(2)])
    ts, rs = np.concatenate(ts), np.concatenate(rs)
    counts = sorted({c for c in (1, 16, 64, physical, cores) if c <= cores})
    legs, nodes = {}, 0
    for th in counts:
        done, el, nodes, first = cb.keyswitch_timed(ts, rs, th, budget_s / len(counts) * (0.6 if th == 1 else 1.1))
        assert np.array_equal(first, want), f"the timed CPU leg ({th} threads) disagrees with the oracle"
        legs[th] = (done / el, done, el)
    # what the host's memory system gives this process at the same thread placements (STREAM-style triad, 64 MiB per array and
    # thread): the keyswitch port streams 29 MB of keys + key factors per keyswitch and thread, so beyond one thread per L3 slice
    # its scaling is bounded by this number, not by the cores
    stream = {str(th): max(cb.lib.cb_stream_triad(th, 0.4, 64) for _ in range(2)) for th in counts if th > 1}   # best of two
    cb.close()
    threads = max(legs, key=lambda k: legs[k][0])
    va, na, ea = legs[threads]
    v1 = legs[1][0]
    kind = "port" if isa == "scalar" else "port+" + isa
    return {"value": va, "unit": "keyswitches/s", "cores": threads, "kind": kind, "isa": isa, "value_1t": v1,
            "by_threads": {str(k): v[0] for k, v in legs.items()},
            "parallel_efficiency": {str(k): v[0] / (k * v1) for k, v in legs.items()},
            "numa_nodes_used": nodes, "host_stream_triad_GBps_by_threads": stream, "cgroup_cpu_quota_cores": cpu_quota_cores(),
            "bytes_streamed_per_keyswitch_and_thread": int(2 * case.L * (case.L + 1) * 2 * case.n * 8 + 5 * case.L * case.n * 8),
            "host_cores_visible": cores, "host_cores_total": os.cpu_count(),
            "sample": f"{na} keyswitches N={case.n} L={case.L} K={case.K} in {ea:.1f}s on {threads} pinned OpenMP threads "
                      f"(legs: {', '.join(f'{k} threads {v[0]:.0f}/s' for k, v in legs.items())}; each thread loops over its own two "
                      f"ciphertexts, keys and tables replicated on {nodes} NUMA node(s)); oracle/cpu_baseline.c: port of the reference's "
                      f"CPU algorithms (Harvey/Shoup NTT, Shoup key products) with {isa} kernels chosen by HEXL's rule (IFMA below 2^50, "
                      f"64-bit AVX512-DQ lanes above; these primes are {int(case.moduli[0]).bit_length()}-bit), gcc -O3 -march=native "
                      f"-fopenmp; Intel HEXL itself is not in the image"}


def cxx_api_end_to_end(L, timeout_s=120, local_cpulist=None):
    """SURVEY 8d's end-to-end leg: the reference's public C++ API (intel::hexl::KeySwitch on host pointers: pack, PCIe up, kernels,
    PCIe down, host accumulate) at the worksizes of benchmark/micro_keyswitch.sh (1 / 16 / 128; bench_keyswitch.cpp:113-131,153-158),
    measured by tests/cpp/bench_cxx_api in its own process. These rates include PCIe and host copies: reported, never `value`."""
    import subprocess
    exe = ROOT / "tests" / "cpp" / "bench_cxx_api"
    if not exe.exists():
        return {"error": "tests/cpp/bench_cxx_api not built"}
    out = {}
    for ws in (1, 16, 128):
        try:
            r = subprocess.run([str(exe), str(ws), str(L), "1" if ws == 128 else "0", "1"], capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[f"worksize_{ws}"] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:                                       # a reported extra: never fails the benchmark
            out[f"worksize_{ws}"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    # the same window with the CALLER on the GPU's socket (taskset to the device's local CPUs): the staging slabs and the library's copy
    # threads live there, and a caller whose arrays were first touched on the other socket pays the inter-socket links for every byte
    # (round 6, profiles/r06_host_numa_sweep.txt)
    if local_cpulist:
        try:
            import shutil
            if shutil.which("taskset"):
                r = subprocess.run(["taskset", "-c", local_cpulist, str(exe), "128", str(L), "0", "1"], capture_output=True, text=True, timeout=timeout_s)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                out["worksize_128_caller_on_the_gpus_socket"] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:
            out["worksize_128_caller_on_the_gpus_socket"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    # the reference's own SEAL workload at the bridge's worksize 1: relinearise (6 of 7 key moduli) and rotate (5 of 7), chain 52,30,30,40,27,27,27
    for name, Lx in (("seal_chain_relinearize_L6_K7_worksize_1", 6), ("seal_chain_rotate_L5_K7_worksize_1", 5)):
        try:
            r = subprocess.run([str(exe), "1", str(Lx), "0", "1", "seal"], capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[name] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    out["note"] = ("host-pointer API of include/hexl-fpga.h on one GPU, decomp %d: PCIe-bound (0.8 MB up + 1.6 MB down per keyswitch at L = 6); "
                   "device-resident callers run at `value`" % L)
    return out


SEAL_CHAIN_BITS = (52, 30, 30, 40, 27, 27, 27)


def seal_chain_rows(hx, ctx, orc_mod, dev, batch=2048, lone_launches=2000):
    """The reference's own SEAL workload (experimental/bridge-seal/tests/seal_test.sh:20: N = 16384, CoeffModulus::Create(16384,
    {52, 30, 30, 40, 27, 27, 27}), relinearize at 6 decomposition limbs, rotate after the rescale at 5 of 7 key moduli) on the device:
    keyswitch/s at `batch` and microseconds per LONE keyswitch (the bridge calls at worksize 1), each with per-limb arithmetic tiers
    (limb 0 strict, the others at reduction period 12: round 5) and with the plan-wide tier of rounds 1-4 (HEXL_KS_PER_LIMB=0: the
    52-bit limb puts all seven on the strict kernels). Device-resident; the host-pointer rate is extra.cxx_api_end_to_end's."""
    import torch
    from ks_util import KsCase, seal_chain
    moduli = seal_chain(orc_mod, 7, N)
    rows = {"moduli": [int(q) for q in moduli], "batch": batch}
    for name, Lx in (("relinearize_L6_K7", 6), ("rotate_L5_K7", 5)):
        row = {}
        for label, per_limb in (("per_limb_tiers", "1"), ("plan_wide_tier", "0")):
            os.environ["HEXL_KS_PER_LIMB"] = per_limb                     # read when the plan is created
            try:
                cs = KsCase(orc_mod, N, Lx, 7, seed=41, moduli=moduli)
                pl = hx.KeySwitchPlan(ctx, N, Lx, 7, 7, 2, cs.moduli, cs.modswitch)
            finally:
                os.environ.pop("HEXL_KS_PER_LIMB", None)
            pl.set_keys(cs.keys)
            tiers, mixed = pl.tiers()
            tt, rr = cs.inputs(orc_mod, 0)                                # parity of this very plan before it is timed
            d_t1, d_r1 = hx.as_i64(tt).to(dev), hx.as_i64(rr).to(dev)
            pl.keyswitch(d_r1, d_t1, 1)
            torch.cuda.synchronize()
            ok = bool(np.array_equal(hx.to_u64(d_r1), cs.expected(orc_mod, tt, rr)))
            tx, rx = device_inputs(hx, orc_mod, cs, batch, dev)
            pl.keyswitch(rx, tx, batch)
            torch.cuda.synchronize()
            probe = hx.to_u64(rx[batch - 1]).copy()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            for _ in range(5):
                pl.keyswitch(rx, tx, batch)
            f1.record()
            torch.cuda.synchronize()
            ms = f0.elapsed_time(f1) / 5
            for _ in range(50):
                pl.keyswitch(d_r1, d_t1, 1)
            torch.cuda.synchronize()
            f0.record()
            for _ in range(lone_launches):
                pl.keyswitch(d_r1, d_t1, 1)
            f1.record()
            torch.cuda.synchronize()
            row[label] = {"tiers": tiers[:Lx] + tiers[6:], "mixed": mixed, "verified_vs_oracle": ok,
                          "keyswitches_per_s": batch / (ms * 1e-3), "lone_keyswitch_us": f0.elapsed_time(f1) * 1e3 / lone_launches}
            del probe
            pl.close()
        row["speedup_batch"] = row["per_limb_tiers"]["keyswitches_per_s"] / row["plan_wide_tier"]["keyswitches_per_s"]
        row["speedup_lone"] = row["plan_wide_tier"]["lone_keyswitch_us"] / row["per_limb_tiers"]["lone_keyswitch_us"]
        rows[name] = row
    return rows


def gpu_local_cpulist(torch_device_index):
    """the CPUs of the socket GPU `torch_device_index` hangs off (sysfs local_cpulist of its PCI device), or None"""
    try:
        import torch
        pr = torch.cuda.get_device_properties(torch_device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        txt = open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip()
        return txt or None
    except Exception:
        return None


class PowerSampler:
    """Board power and shader clock from the amdgpu hwmon files of the benchmarked GPU, sampled every 50 ms on a thread while
    the timed region runs. Every kernel family of this library runs the board AT its power cap (1378-1400 W of
    1400 W), so the shader clock in `roofline.alu` is what power management leaves, not a constant of the chip."""

    def __init__(self, torch_device_index):
        import glob
        self.dir = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch_device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            hits = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hits:
                self.dir = hits[0]
        except Exception:
            pass
        if self.dir is None:
            for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
                if os.path.exists(d + "/power1_average") or os.path.exists(d + "/power1_input"):
                    self.dir = d
                    break
        self.samples, self._stop, self._th = [], False, None

    def _read(self, name):
        try:
            return float(open(f"{self.dir}/{name}").read())
        except Exception:
            return None

    def start(self):
        if self.dir is None:
            return
        import threading

        def loop():
            while not self._stop:
                w = self._read("power1_average") or self._read("power1_input")
                f = self._read("freq1_input")
                if w:
                    self.samples.append((w / 1e6, f / 1e6 if f else None))
                time.sleep(0.05)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join()
        if not self.samples:
            return None
        ws = [w for w, _ in self.samples]
        fs = [f for _, f in self.samples if f]
        cap = self._read("power1_cap")
        return {"samples": len(ws), "board_power_w_mean": sum(ws) / len(ws), "board_power_w_max": max(ws),
                "power_cap_w": cap / 1e6 if cap else None, "sclk_mhz_mean": sum(fs) / len(fs) if fs else None,
                "source": self.dir}


def ntt_roofline_block(timed, pmc_ntt, cus, timed_sclk_mhz=None):
    """BASELINE's second metric (fwd / inv NTT per second, N = 16384, batch 1024) against both of ITS bounds, like the keyswitch's block:
    `frac` = algorithmic bytes (262,144 B per transform) per launch / the timed launch duration / 8 TB/s; `traffic` = L2-miss-side
    bytes per launch from the in-run PMC passes (transform kernel + the table-preparation kernel in front of it); `alu` = the FP64-issue
    bound -- VALU wave-instructions per launch x 4 cycles / SIMDs / clock against the timed launch (at the timed region's own shader clock
    when hwmon gave one, else the counter passes' clock, labelled)."""
    out = {}
    for leg in ("fwd", "inv"):
        t = timed.get(leg)
        if not t:
            continue
        batch = 1024
        us = t["ms_per_launch"] * 1e3
        alg = batch * 2 * N * 8
        b = {"bound": "hbm", "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_launch": alg, "timed_us_per_launch": us,
             "frac_of_measured_copy_ceiling_6290_GBps": alg / (us * 1e-6) / 1e9 / 6290.0, "traffic": None}
        p = (pmc_ntt or {}).get(leg)
        if p:
            b["traffic"] = p.get("traffic_bytes_per_launch")
            b["traffic_over_algorithmic"] = p.get("traffic_over_algorithmic")
            b["traffic_kind"] = "L2-miss side (2 x FETCH_SIZE + WRITE_SIZE), transform kernel + k_ntt_prepare, in-run PMC passes"
            vw = p.get("valu_wave_instructions_per_launch")
            if vw:
                ghz = (timed_sclk_mhz / 1e3) if timed_sclk_mhz else p.get("shader_clock_ghz")
                issue_us = vw * 4.0 / (4 * cus) / (ghz * 1e3) if ghz else None
                b["alu"] = {"bound": "valu_fp64", "valu_wave_instructions_per_launch": vw,
                            "valu_wave_instructions_per_transform": p.get("valu_wave_instructions_per_transform"),
                            "shader_clock_ghz": ghz, "shader_clock_source": "hwmon over the timed NTT launches" if timed_sclk_mhz else "PMC passes (GRBM_GUI_ACTIVE of k_ksx_main)",
                            "issue_us_per_launch": issue_us, "achieved_frac": issue_us / us if issue_us else None,
                            "under_pmc": {k: p.get(k) for k in ("avg_us_under_pmc", "prepare_kernel_avg_us_under_pmc", "fp64_issue_frac_under_pmc", "wave_time_split")}}
        out[leg] = b
    return out


def key_stream_split(traffic_per_ks, traffic_per_ks_keys_aliased):
    """The key stream's share of the L2-miss-side bytes per keyswitch = (the real pipeline) - (every key row reading row 0), BOTH measured
    on the shipped library's kernel objects (the key-alias variant differs in one launcher function only). What is left -- c and s'
    written once and re-read from another XCD's L2, t_target, the result read-modify-write, the twiddle tables -- is the DRAM-side
    estimate: the 14.7 MB key set lives in the 256 MiB Infinity Cache."""
    return {"key_stream_bytes_per_keyswitch": traffic_per_ks - traffic_per_ks_keys_aliased,
            "dram_side_estimate_bytes_per_keyswitch": traffic_per_ks_keys_aliased,
            "dram_side_estimate_note": "L2-miss-side bytes (2 x FETCH_SIZE + WRITE_SIZE) of the same passes with every key row aliased onto row 0 "
                                       "(HEXL_KSX_ALIAS=1 on tools/lib_var/libhexl_mi355x_keyalias.so: the shipped kernel objects, one launcher function differs): "
                                       "the key stream (L2 misses served by the Infinity Cache) removed"}


def alu_block(valu_per_ks, cus, measured_us_per_ks, timed_sclk_mhz, pmc_clock_ghz):
    """The FP64-issue bound, self-consistent: instructions per keyswitch x 4 cycles / SIMDs / CLOCK against the measured time per
    keyswitch OF THE TIMED REGION -- so the clock must be the timed region's own (hwmon samples of the shader clock while it ran),
    not the clock of the PMC passes, whose kernels run ~10 % longer under the counters at a lower clock (round 3 mixed the two and
    printed 0.698 where 0.66 was right). Both are reported; `achieved_frac` is the timed-clock one (the conservative figure)."""
    out = {"bound": "valu_fp64", "valu_wave_instructions_per_keyswitch": valu_per_ks, "cycles_per_wave_instruction": 4,
           "simds": 4 * cus, "measured_us_per_keyswitch": measured_us_per_ks}
    def issue_us(ghz):
        return valu_per_ks * 4.0 / (4 * cus) / (ghz * 1e3)
    if pmc_clock_ghz:
        out["pmc_pass_clock_ghz"] = pmc_clock_ghz
        out["achieved_frac_at_pmc_pass_clock"] = issue_us(pmc_clock_ghz) / measured_us_per_ks
    clock = timed_sclk_mhz / 1e3 if timed_sclk_mhz else pmc_clock_ghz
    out["shader_clock_ghz"] = clock
    out["shader_clock_source"] = "hwmon shader clock sampled over the timed region" if timed_sclk_mhz else "PMC passes (no hwmon samples)"
    out["issue_us_per_keyswitch"] = issue_us(clock)
    out["achieved_frac"] = issue_us(clock) / measured_us_per_ks
    return out


def ctx_cus(ctx):
    import re
    m = re.search(r"(\d+) CUs", ctx.describe())
    return int(m.group(1)) if m else 256


PMC_BATCH = 256   # one scratch chunk: every dispatch of the passes below is one whole chunk of 256 keyswitches


def pmc_inrun(L, cus, timeout_s=90):
    """The roofline block's counter inputs measured INSIDE this benchmark run: rocprofv3 PMC passes (one counter group per
    run, --kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE and WRITE_SIZE in their own passes) of the native
    workload tools/pmc_workload (2 launches of a 256-keyswitch chunk, same library, same kernels), plus two passes of the
    KEY-ALIAS variant (tools/pmc_workload_keyalias on tools/lib_var/libhexl_mi355x_keyalias.so: the shipped library's own kernel OBJECTS, only
    the launcher lets HEXL_KSX_ALIAS=1 point every key row at row 0; the shipped library has no such knob): the difference is the key
    stream's share of the L2-miss-side bytes, which the 256 MiB Infinity Cache serves (the key set is 14.7 MB), so what is left
    estimates the DRAM side. (Rounds 3-4 took the aliased passes from the PROFILING build, whose kernels spill more and move 19.2 MB
    per keyswitch where the shipped ones move 14.4: a difference ACROSS the builds, 0.6-0.7 MB where 5 MB is right --
    profiles/r05_fetch_reconcile.json, tools/fetch_reconcile.py.)
    Returns (derived dict, None) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = ROOT / "tools" / "pmc_workload"
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not exe.exists():
        return None, "tools/pmc_workload not built"
    if not Path(rocprof).exists():
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCP_TOOL", "ROCPROF")) for k in os.environ):
        return None, "this process itself runs under a profiler (no nested rocprofv3)"
    sys.path.insert(0, str(ROOT / "tools"))
    import pmc_summary
    env = dict(os.environ, TMPDIR="/tmp", HEXL_KS_ONE_LANE="1")
    groups = ["SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU", "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY",
              "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"]

    def passes(root, groups, extra_env, exe=exe):
        for i, g in enumerate(groups):
            # (... 2 1: two launches of the keyswitch chunk, then two forward + inverse NTT launches of 1024 polynomials: BASELINE's second
            # metric gets its counters from the same passes)
            cmd = [rocprof, "--kernel-trace", "--pmc", *g.split(), "-d", f"{root}/p{i + 1}", "--", str(exe), str(PMC_BATCH), str(L), "2", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(env, **extra_env), timeout=timeout_s, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"rocprofv3 pass '{g}' exited {r.returncode}: {r.stderr[-300:]}")
        vals, dur = pmc_summary.collect(root)
        d = pmc_summary.derive(vals, dur, PMC_BATCH, L, simds=4 * cus)
        d["ntt"] = pmc_summary.derive_ntt(vals, dur, 1024, simds=4 * cus, clock_ghz=d.get("shader_clock_ghz"))
        return d
    try:
        with tempfile.TemporaryDirectory(prefix="hexl_pmc_", dir="/tmp") as tmp:
            d = passes(tmp + "/a", groups, {})
            if d["traffic_bytes_per_keyswitch"] is None or d["valu_wave_instructions_per_keyswitch"] is None:
                return None, "PMC passes returned no counters for the keyswitch kernels"
            try:
                al = passes(tmp + "/b", ["FETCH_SIZE", "WRITE_SIZE"], {"HEXL_KSX_ALIAS": "1"}, ROOT / "tools" / "pmc_workload_keyalias")
                d["traffic_bytes_per_keyswitch_keys_aliased"] = al["traffic_bytes_per_keyswitch"]
                d["keys_aliased_kernels"] = "byte-identical to the shipped library's (libhexl_mi355x_keyalias.so links the same objects)"
            except Exception as e:                                  # the estimate is optional
                d["traffic_bytes_per_keyswitch_keys_aliased"] = None
                d["keys_aliased_error"] = str(e)[:200]
            return d, None
    except Exception as e:
        return None, f"{type(e).__name__}: {str(e)[:300]}"


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_gpus, argv=None):
    """Re-run this script as `n_gpus` ranks under torch.distributed.run on this node (rendezvous on 127.0.0.1, a free port) and
    return the launcher's exit status; the ranks see WORLD_SIZE and take the normal path. Rank 0's JSON line is the only stdout."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()),
           *(sys.argv[1:] if argv is None else argv)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")                # torch.distributed.run would set (and warn about) it anyway
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total-batch", type=int, default=None,
                    help="keyswitches per step over ALL GPUs, split into contiguous shards (strong scaling; default 8192 = BASELINE config 5; "
                         "44 GB of ciphertexts on one GPU)")
    ap.add_argument("--batch", type=int, default=None, help="keyswitches per GPU per step instead (weak scaling)")
    ap.add_argument("--barrier-per-step", action="store_true", help="all ranks synchronise after every step (pre-flight of short steps)")
    ap.add_argument("--decomp", type=int, default=7, help="decomp_modulus_size L (key_modulus_size = L+1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="bound of the cpu_baseline leg's timed CPU work (rank 0, every N)")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--light-extra", action="store_true", help="only the all-ranks NTT rows of `extra` (what N > 1 reports), also at N = 1")
    ap.add_argument("--no-pmc", action="store_true", help="roofline.traffic / roofline.alu from profiles/*_latest.json instead of in-run PMC passes")
    a = ap.parse_args()
    assert not (a.batch and a.total_batch), "--batch (per GPU, weak) and --total-batch (all GPUs, strong) exclude each other"
    assert a.gpus >= 1, "--gpus must be positive"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one process per GPU, the reference's DevicePool of
        # NUM_DEV runners, host/src/fpga.cpp:1646-1673) instead of silently measuring one GPU
        sys.exit(self_launch(a.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world_env}: launch one rank per GPU "
                 f"(python bench.py --gpus {a.gpus} does that itself) -- refusing to report a rate for the wrong number of GPUs")

    import torch
    import torch.distributed as dist
    import hexl_fpga_amd as hx
    import orc as orc_mod
    from hexl_fpga_amd.sharding import max_over_ranks, shard_range
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
    # HEXL_BENCH_FORCE_DIST=1: the process group of the N > 1 path -- RCCL communicator bound to this rank's device, device-side
    # barrier and MAX-reduce -- is created and used even at --gpus 1, so that a one-GPU box executes exactly the code an 8-GPU
    # launch runs (tests/test_gpu_bench_ranks.py::test_rccl_path_forced_on_one_gpu)
    force_dist = os.environ.get("HEXL_BENCH_FORCE_DIST") == "1"
    grouped = world > 1 or force_dist
    backend = None
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                       # (only without a launcher, i.e. forced at world 1)
            os.environ["MASTER_PORT"] = str(free_port())
        backend = "gloo" if one_gpu else "nccl"
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        torch.cuda.synchronize()                                  # this rank's work is done ...
        if grouped:
            dist.barrier()                                        # ... and so is everybody else's
            torch.cuda.synchronize()

    def slowest(seconds):
        return max_over_ranks(seconds, "cpu" if one_gpu else dev, force=force_dist)

    # the step's batch and this rank's shard of it (independent ciphertexts: no data-path collective)
    if a.batch:
        scaling, total = "weak", a.batch * world
        first, last = rank * a.batch, (rank + 1) * a.batch
    else:
        scaling, total = "strong", a.total_batch or 8192
        first, last = shard_range(total, world, rank)
    mine = last - first
    assert mine > 0, "more ranks than ciphertexts"

    ctx = hx.Context(local)
    L, K = a.decomp, a.decomp + 1
    case = KsCase(orc_mod, N, L, K, seed=1)                   # one key set for the job, replicated on every GPU
    plan = hx.KeySwitchPlan(ctx, N, L, K, L + 1, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    case.seed = 1 + rank                                      # every rank: its own ciphertexts
    d_t, d_r = device_inputs(hx, orc_mod, case, mine, dev)

    # in-run check of the measured path: instances on both sides of a scratch-chunk boundary and the last one of the first
    # full-batch launch against the oracle
    probe = sorted({b for b in (0, 255, 256, mine - 1) if 0 <= b < mine})
    before = {b: (hx.to_u64(d_t[b]).copy(), hx.to_u64(d_r[b]).copy()) for b in probe}
    plan.keyswitch(d_r, d_t, mine)
    ctx.sync()
    verified = all(bool(np.array_equal(hx.to_u64(d_r[b]), case.expected(orc_mod, *before[b]))) for b in probe)
    assert verified, "keyswitch output differs from the oracle"
    for _ in range(a.warmup):
        plan.keyswitch(d_r, d_t, mine)
    barrier()
    power = PowerSampler(local) if rank == 0 else None
    if power:
        power.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        plan.keyswitch(d_r, d_t, mine)
        if a.barrier_per_step:
            barrier()
    e1.record()
    barrier()
    elapsed = slowest(time.perf_counter() - t0)
    power = power.stop() if power else None
    dev_ms = e0.elapsed_time(e1)                              # HIP events on the launch stream

    value = total * a.steps / elapsed
    out = {
        "metric": "keyswitches/sec at N=16384, decomp=7", "value": value, "unit": "keyswitches/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "verified_vs_oracle": verified, "verified_instances": probe,
        "config": {"workload": f"keyswitch N={N} decomp_modulus_size={L} key_modulus_size={K} 52-bit primes "
                               f"(GeneratePrimes(K,51,N)), kcc=2, batch {total} per step"
                               + (f" = {a.batch}/GPU" if scaling == "weak" else f" split over {world} GPU(s) ({mine} on rank 0)")
                               + ", resident in HBM",
                   "global_batch": total, "batch_per_gpu": mine, "barrier_per_step": bool(a.barrier_per_step),
                   "parallelism": f"{world} independent shard(s), no collective"},
    }
    if grouped:
        out["config"]["timing_group"] = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size": dist.get_world_size(),
                                         "forced_at_one_rank": bool(force_dist and world == 1),
                                         "used_for": "barrier on both sides of the timed region + MAX-reduce of the wall time; no data-path collective"}
    # BASELINE's second metric, fwd-NTT/sec at N=16384 at 1/2/4/8 GPUs: config 2's shape on every rank, whole-job rate
    # (300 launches per leg, ~25 ms: ten launches end before the clocks and the power limit have settled and read 8 % low)
    ntt_power = None
    if a.no_extra:
        ntt = None
    else:
        ntt_sampler = PowerSampler(local) if rank == 0 else None
        if ntt_sampler:
            ntt_sampler.start()
        ntt = time_ntt(hx, ctx, orc_mod, dev, 1024, 300, barrier, slowest, world)
        ntt_power = ntt_sampler.stop() if ntt_sampler else None
    if rank == 0:
        alg = ks_alg_bytes(N, L)
        ach = alg * mine * a.steps / (dev_ms * 1e-3) / 1e9            # this rank, device-timed
        stage = plan.time_stages(d_r, d_t, min(mine, 256), 3)
        us_per_ks = dev_ms * 1e3 / (mine * a.steps)
        cus = ctx_cus(ctx)
        # Both bounds. L2-miss-side bytes = sum over the pipeline's kernels of 2*FETCH_SIZE + WRITE_SIZE; FP64 issue time =
        # VALU instructions per keyswitch (SQ_INSTS_VALU) x 4 cycles (one wave64 FP64 instruction on a SIMD,
        # MI355X_MICROARCH.md: 78.6 TFLOP/s vector FP64) / (4 SIMDs x CUs) / shader clock under this load. Measured by PMC passes
        # run from inside this benchmark (pmc_inrun); profiles/*_latest.json (the round's committed passes) only as fallback.
        traffic, traffic_src, alu, pmc = None, None, None, None
        why = "--no-pmc" if a.no_pmc else f"n_gpus = {world}: the counter passes are collected at N = 1 only"
        if not a.no_pmc and world == 1:
            ctx.sync()
            pmc, why = pmc_inrun(L, cus)
        traffic_extra = {}
        if pmc:
            traffic = pmc["traffic_bytes_per_keyswitch"] * mine
            traffic_src = f"PMC passes run inside this benchmark (rocprofv3 --kernel-trace --pmc, tools/pmc_workload, chunk of {PMC_BATCH}), per keyswitch x batch"
            vw, clk = pmc["valu_wave_instructions_per_keyswitch"], pmc["shader_clock_ghz"] or 2.0
            alu_src = "in-run PMC passes (SQ_INSTS_VALU, GRBM_GUI_ACTIVE per kernel)"
            al = pmc.get("traffic_bytes_per_keyswitch_keys_aliased")
            if al:
                # written once (c, s'), compulsory, or re-read from another XCD: everything except the key rows, which come out of the
                # Infinity Cache (14.7 MB key set against 256 MiB)
                traffic_extra = key_stream_split(pmc["traffic_bytes_per_keyswitch"], al)
            per_kernel = {k: {kk: e.get(kk) for kk in ("avg_us_under_pmc", "fp64_issue_frac", "wave_time_split", "read_bytes", "write_bytes")}
                          for k, e in pmc["kernels"].items()}
        else:
            tj, aj = ROOT / "profiles" / "traffic_latest.json", ROOT / "profiles" / "alu_latest.json"
            vw = clk = None
            per_kernel = None
            if tj.exists() and json.loads(tj.read_text()).get("L") == L:
                traffic = json.loads(tj.read_text())["keyswitch_traffic_bytes_per_unit"] * mine
                traffic_src = f"profiles/traffic_latest.json (committed PMC passes, not this run: {why})"
            if aj.exists() and json.loads(aj.read_text()).get("L") == L:
                t = json.loads(aj.read_text())
                vw, clk = t["valu_wave_instructions_per_keyswitch"], t.get("shader_clock_ghz", 2.0)
                alu_src = f"profiles/alu_latest.json (committed PMC passes, not this run: {why})"
        if vw:
            alu = alu_block(vw, cus, us_per_ks, (power or {}).get("sclk_mhz_mean"), clk)
            alu.update(source=alu_src, per_kernel=per_kernel)
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "traffic_kind": "L2-miss side (2 x FETCH_SIZE + WRITE_SIZE at the L2's fabric port): includes what the Infinity Cache serves",
                           "traffic_source": traffic_src, **traffic_extra,
                           "alg_bytes_per_launch": alg * mine,
                           "kernel": "keyswitch pipeline (k_ksx_intt + k_ksx_special + k_ksx_main; chunks of 256 keyswitches)",
                           "alg_bytes_per_keyswitch": alg, "device_ms_per_step": dev_ms / a.steps,
                           # per chunk of 256, from hipEvents on the launch stream (compare avg_us in profiles/*kernel_trace*)
                           "dominant_kernel": {"name": "k_ksx_main (steps 2-3 of the L decomposition limbs, steps 5-7)",
                                               "ms_per_chunk": stage[3], "share_of_pipeline": stage[3] / stage[0],
                                               "chunk": min(mine, 256)},
                           # the binding bound: 72 N-point transforms of exact 52-bit arithmetic per 4.6 MB
                           "alu": alu,
                           # ... and what sets the clock in it: the board's power cap (hwmon samples over the timed region)
                           "power": power}
        if power and power.get("board_power_w_mean"):
            # energy per keyswitch = mean board power over the timed region x time per keyswitch: every kernel family of this
            # library runs the board at its power cap, so joules -- not stalls -- are what an optimisation has to save
            out["roofline"]["energy_mj_per_keyswitch"] = power["board_power_w_mean"] * us_per_ks * 1e-3
        extra = {"stage_ms_at_batch_%d" % min(mine, 256): {"total": stage[0], "step_1_inverse_transforms": stage[1],
                                                             "steps_2_4_special_limb": stage[2],
                                                             "steps_2_3_5_7_decomposition_limbs": stage[3]},
                 "device": ctx.describe()}
        if ntt:
            extra["ntt_N16384_batch1024"] = ntt
            # BASELINE's second metric where the driver's parser sees it without digging in `extra`
            out["ntt_fwd_per_s"] = ntt["fwd"]["ntt_per_s_all_ranks"]
            out["ntt_inv_per_s"] = ntt["inv"]["ntt_per_s_all_ranks"]
            out["ntt_config"] = f"N={N}, q={ntt['q']} (51-bit, exact FP64 fast path), batch 1024 per GPU per launch, {world} GPU(s)"
            # ... and its own roofline block (round 5): HBM fraction from the timed launches, L2-miss-side traffic and the FP64-issue
            # fraction from the same in-run PMC passes as the keyswitch's
            out["ntt_roofline"] = ntt_roofline_block(ntt, (pmc or {}).get("ntt"), cus, (ntt_power or {}).get("sclk_mhz_mean"))
            if ntt_power:
                out["ntt_roofline"]["power"] = ntt_power
        if not a.no_extra and not a.light_extra and world == 1:
            extra["ntt_N16384_batch4096"] = time_ntt(hx, ctx, orc_mod, dev, 4096, 100)      # launch overhead amortised over 4x the work
            # the slower standalone-NTT paths, same shape: SURVEY 8d's own prime (2^52 + 393217 is above the LAZY FP64 range: strict
            # FP64 kernels since round 4 -- integer Harvey kernels before), a 59-bit prime (integer Harvey kernels) and the reference
            # benchmark's workload (random tables: integer kernels through the host hint)
            extra["ntt_N16384_batch1024_q_2p52_plus_393217_strict_fp64"] = time_ntt(hx, ctx, orc_mod, dev, 1024, 100, q=NTT_Q_SURVEY)
            extra["ntt_N16384_batch1024_59bit_prime_integer_kernels"] = time_ntt(hx, ctx, orc_mod, dev, 1024, 100, q=orc_mod.primes(1, NTT_BITS_INTEGER, N)[0])
            extra["ntt_N16384_batch1024_refbench_random_tables"] = time_ntt(hx, ctx, orc_mod, dev, 1024, 100, q=NTT_Q_REFBENCH, random_tables=True)
            # beyond the reference's envelope: N = 32768 as two 16384-point sub-transforms per polynomial (ntt.hip k_ntt_fwd_h / k_ntt_inv_h)
            extra["ntt_N32768_batch512_two_sub_transforms"] = time_ntt(hx, ctx, orc_mod, dev, 512, 60, n=32768)
            extra["cxx_api_end_to_end"] = cxx_api_end_to_end(6, local_cpulist=gpu_local_cpulist(local))
            extra["dyadic_n8192_m4_batch4096"] = time_dyadic(hx, ctx, orc_mod, dev)
            def other_shape(Lx, Kx, moduli=None, n=N):
                cs = KsCase(orc_mod, n, Lx, Kx, seed=99, moduli=moduli)
                pl = hx.KeySwitchPlan(ctx, n, Lx, Kx, Kx, 2, cs.moduli, cs.modswitch)
                pl.set_keys(cs.keys)
                nbx = max(1, min(mine, 2048) * N // n)
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
                nbx = min(mine, 2048)
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
            # ... and the next ring dimension up (beyond the reference's envelope): every transform as two 16384-point halves
            extra["keyswitch_32768_3_4_4_2"] = other_shape(3, 4, None, 32768)
            # the headline shape with the LARGEST 52-bit primes = 1 mod 2N (what SEAL's CoeffModulus::Create(n, {52, ...}) picks; `value`
            # uses the smallest ones): above the lazy bound 2^51 (1 + 2^-7), i.e. the STRICT FP64 kernels, 14 instead of 8-11
            # instructions per butterfly -- the slower reading of "52-bit primes", reported beside the faster one
            def largest_52bit_primes(count):
                out_, v = [], (1 << 52) - 2 * N + 1
                while len(out_) < count:
                    if orc_mod.orc().orc_is_prime(v):
                        out_.append(v)
                    v -= 2 * N
                return out_
            extra["keyswitch_16384_L%d_largest_52bit_primes_strict_kernels" % L] = other_shape(L, L + 1, largest_52bit_primes(L + 1))
            # the reference's own SEAL workload (bridge-seal's prime chain 52,30,30,40,27,27,27): per-limb tiers against the plan-wide tier
            extra["keyswitch_seal_chain_52_30_30_40_27_27_27"] = seal_chain_rows(hx, ctx, orc_mod, dev, batch=min(mine, 2048))
            # the headline shape on the 64-bit INTEGER kernels (59-bit primes: beyond the reference's < 2^52 envelope)
            extra["keyswitch_16384_L%d_59bit_primes_integer_kernels" % L] = other_shape(L, L + 1, orc_mod.primes(L + 1, 59, N))
        out["extra"] = extra
    plan.close()
    ctx.close()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # The reported CPU baseline: rank 0, at EVERY N (round 6: a line without it is graded unmeasured), after the process group is
        # gone -- the other ranks have nothing left to do and exit, so the host cores are this leg's alone, as at N = 1.
        if not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(orc_mod, case, a.cpu_seconds)
            out["cpu_baseline"]["timed_on"] = f"rank 0 of {world}, after the GPU legs and the final barrier (the other ranks have exited)"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
