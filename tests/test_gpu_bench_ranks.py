"""GPU: the N > 1 path of bench.py executed the way the driver launches it (torch.distributed.run, one process per rank,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) -- on the one GPU this box has: HEXL_BENCH_ONE_GPU=1 puts
both ranks on GPU 0 and routes the timing barrier and the max-reduce through gloo. What it pins: BASELINE config 5's shape
(ONE batch per step split into contiguous shards, `scaling: "strong"`; 2048 over two ranks = the 1024 per GPU of 8192 over
eight), every rank passing the in-run oracle check on both sides of a scratch-chunk boundary, rank 0 alone printing ONE JSON
line whose value is the whole-job rate, the all-ranks NTT rate (BASELINE's second metric), and the weak mode of --batch."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_ranks(port, *args):
    env = dict(os.environ, HEXL_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu", *args]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    return json.loads(lines[0])


def test_two_ranks_strong_shape_of_config_5():
    d = run_ranks(29517, "--total-batch", "2048", "--barrier-per-step")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["verified_vs_oracle"] is True
    assert d["config"]["global_batch"] == 2048 and d["config"]["batch_per_gpu"] == 1024      # 1024 per rank: config 5's shard
    assert d["verified_instances"] == [0, 255, 256, 1023]
    assert d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    # whole-job aggregate: 2048 keyswitches x 3 steps over the slowest rank's wall time
    assert abs(d["value"] - 2048 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"
    ntt = d["extra"]["ntt_N16384_batch1024"]
    for leg in ("fwd", "inv"):                                   # BASELINE's second metric at N ranks: all ranks' transforms
        assert ntt[leg]["n_gpus"] == 2 and ntt[leg]["ntt_per_s_all_ranks"] > 0


def test_two_ranks_weak_mode():
    d = run_ranks(29519, "--batch", "512", "--no-extra")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["verified_vs_oracle"] is True
    assert d["config"]["global_batch"] == 1024 and d["config"]["batch_per_gpu"] == 512
    assert abs(d["value"] - 2 * 512 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_gpus_flag_alone_launches_the_ranks():
    """`python bench.py --gpus 2` with NO external launcher (the shape of the driver's N = 1 command with another N): bench.py
    launches its own ranks (round 4: the flag was parsed and ignored, a bare `--gpus 8` measured one GPU and printed n_gpus 1)."""
    env = dict(os.environ, HEXL_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu", "--total-batch", "2048"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["verified_vs_oracle"] is True
    assert d["config"]["batch_per_gpu"] == 1024
    ntt = d["extra"]["ntt_N16384_batch1024"]
    assert ntt["fwd"]["n_gpus"] == 2 and ntt["fwd"]["ntt_per_s_all_ranks"] > 0
    # and a launcher whose world disagrees with the flag is refused, not mis-reported
    bad = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=str(ROOT))
    assert bad.returncode != 0 and "refusing" in bad.stderr


def _one_json_line(out):
    print(out.stdout[-3000:], out.stderr[-3000:])
    assert out.returncode == 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    return json.loads(lines[0])


def test_rccl_path_forced_on_one_gpu():
    """The process group of the N > 1 path on real hardware: HEXL_BENCH_FORCE_DIST=1 makes `--gpus 1` create the RCCL communicator
    bound to its device (init_process_group("nccl", device_id=...)), run dist.barrier() on both sides of the timed region and the
    DEVICE-side all_reduce(MAX, float64) of the wall time -- every multi-rank test above goes through gloo because RCCL refuses two
    ranks on one device, so without this the first 8-GPU launch would run that code for the first time."""
    env = dict(os.environ, HEXL_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HEXL_BENCH_ONE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-pmc", "--light-extra",
           "--total-batch", "1024", "--barrier-per-step"]
    d = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT)))
    g = d["config"]["timing_group"]
    assert g["backend"].startswith("nccl") and g["world_size"] == 1 and g["forced_at_one_rank"] is True
    assert d["n_gpus"] == 1 and d["verified_vs_oracle"] is True and d["value"] > 0
    assert abs(d["value"] - 1024 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6   # the reduced wall time IS this rank's
    ntt = d["extra"]["ntt_N16384_batch1024"]
    assert ntt["fwd"]["ntt_per_s_all_ranks"] > 0 and ntt["inv"]["ntt_per_s_all_ranks"] > 0


def test_config_5_as_written_eight_ranks_batch_8192():
    """BASELINE config 5 at its own size: keyswitch N = 16384, decomp 7, batch 8192 sharded over EIGHT ranks (1024 each), launched the
    way the driver launches N = 8 -- all eight on the one GPU of this box (HEXL_BENCH_ONE_GPU=1, timing group over gloo). Every rank
    verifies instances 0 / 255 / 256 / 1023 of ITS shard against the oracle (a rank that fails aborts the launch); the single line
    carries n_gpus 8, `roofline` (counter inputs from the committed passes, labelled) AND `cpu_baseline` (rank 0, after the group is
    gone). The line is kept as profiles/r06_eight_ranks_one_gpu.json."""
    env = dict(os.environ, HEXL_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HEXL_BENCH_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "1", "--cpu-seconds", "8"]
    d = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=str(ROOT)))
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["verified_vs_oracle"] is True
    assert d["config"]["global_batch"] == 8192 and d["config"]["batch_per_gpu"] == 1024
    assert d["verified_instances"] == [0, 255, 256, 1023]
    assert d["config"]["timing_group"]["world_size"] == 8
    assert abs(d["value"] - 8192 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["traffic"] and "committed" in r["traffic_source"]
    assert r["alu"] and "committed" in r["alu"]["source"]
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"].startswith("port") and "rank 0 of 8" in c["timed_on"]
    assert d["extra"]["ntt_N16384_batch1024"]["fwd"]["n_gpus"] == 8 and d["ntt_fwd_per_s"] > 0
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "r06_eight_ranks_one_gpu.json").write_text(json.dumps(d, indent=1))
