#!/usr/bin/env python3
"""Scratch-chunk / lane sizing of the slot-major keyswitch against the 256 MiB Infinity Cache: keyswitch/s, board
power, shader clock and mJ per keyswitch for HEXL_KS_CHUNK x HEXL_KS_LANES, one subprocess per leg (both knobs are read once per
process), legs interleaved `rounds` times. Scratch per instance at L = 7: c[7][n] + s'[2][n] = 1.18 MB, so 2 lanes x 64 = 151 MB
(+ 15 MB of keys + 3.7 MB of tables) fits the Infinity Cache, 2 x 256 = 604 MB (the default) does not.
usage: chunk_energy_sweep.py [batch] [seconds per leg] [rounds] [legs, e.g. 73x2,146x2,256x2] [exact]
"exact" rounds every leg's batch down to a multiple of its chunk (no ragged last chunk)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LEGS = [(256, 2), (128, 2), (96, 2), (64, 2), (64, 3), (64, 4), (128, 4), (512, 2)]


def leg(batch, seconds):
    sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
    import time
    import numpy as np
    import torch
    import hexl_fpga_amd as hx
    import orc
    import bench
    from ks_util import KsCase
    L = 7
    dev = torch.device("cuda:0")
    ctx = hx.Context(0)
    case = KsCase(orc, 16384, L, L + 1, seed=1)
    plan = hx.KeySwitchPlan(ctx, 16384, L, L + 1, L + 1, 2, case.moduli, case.modswitch)
    plan.set_keys(case.keys)
    d_t, d_r = bench.device_inputs(hx, orc, case, batch, dev)
    probe = [0, batch // 2 + 1, batch - 1]
    before = {b: (hx.to_u64(d_t[b]).copy(), hx.to_u64(d_r[b]).copy()) for b in probe}
    plan.keyswitch(d_r, d_t, batch)
    ctx.sync()
    ok = all(np.array_equal(hx.to_u64(d_r[b]), case.expected(orc, *before[b])) for b in probe)
    for _ in range(3):
        plan.keyswitch(d_r, d_t, batch)
    torch.cuda.synchronize()
    ps = bench.PowerSampler(0)
    ps.start()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            plan.keyswitch(d_r, d_t, batch)
        torch.cuda.synchronize()
        reps += 4
    dt = time.perf_counter() - t0
    pw = ps.stop() or {}
    rate = batch * reps / dt
    w = pw.get("board_power_w_mean")
    print(json.dumps({"parity": ok, "keyswitch_per_s": rate, "board_w": w, "sclk_mhz": pw.get("sclk_mhz_mean"),
                      "mj_per_keyswitch": (w / rate * 1e3) if w else None}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--leg":
        leg(int(sys.argv[2]), float(sys.argv[3]))
        sys.exit(0)
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    if len(sys.argv) > 4:
        LEGS = [tuple(int(x) for x in l.split("x")) for l in sys.argv[4].split(",")]
    exact = len(sys.argv) > 5 and sys.argv[5] == "exact"
    print(f"# batch {batch}, {seconds} s per leg, {rounds} interleaved rounds; scratch per lane = chunk x 1.18 MB")
    for r in range(rounds):
        for chunk, lanes in LEGS:
            env = dict(os.environ, HEXL_KS_CHUNK=str(chunk), HEXL_KS_LANES=str(lanes))
            lb = batch // chunk * chunk if exact else batch
            out = subprocess.run([sys.executable, __file__, "--leg", str(lb), str(seconds)], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else {"error": out.stderr[-200:]}
            print(f"round {r} chunk {chunk:4d} lanes {lanes} batch {lb} scratch {chunk * lanes * 1.18:6.0f} MB: " + json.dumps(d), flush=True)
