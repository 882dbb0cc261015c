"""GPU: the N > 1 path of bench.py executed the way the driver launches it (torch.distributed.run, one process per rank,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) -- on the one GPU this box has: HEXL_BENCH_ONE_GPU=1 puts
both ranks on GPU 0 and routes the timing barrier and the max-reduce through gloo. What it pins: every rank builds its own
plan and shard, both pass the in-run oracle check, rank 0 alone prints ONE JSON line whose value is the whole-job rate."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_two_ranks_one_json_line():
    env = dict(os.environ, HEXL_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512",
           "--no-extra", "--no-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["verified_vs_oracle"] is True
    assert d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    # whole-job aggregate: 2 ranks x 512 keyswitches x 3 steps over the slowest rank's wall time
    assert abs(d["value"] - 2 * 512 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"
