#!/usr/bin/env python3
"""trace_scripts.py -- run the REFERENCE's runner scripts (tests/micro_*.sh, benchmark/micro_*.sh) UNMODIFIED, as scripts, in
a directory where every binary they call is a recording stub, and write what they would have executed -- binary, arguments
and the environment each invocation was given -- to _build/script_trace.json.

Why a trace: the scripts are reference sources and cannot travel to the GPU box (nor be copied into this repository), while
the binaries they call need the GPU. So the scripts are executed here, by bash, from where they lie (RUN_CHOICE=1 makes them
skip `aocl initialize`; bitstream_dir.sh is sourced from their own directory), and `pytest -m gpu`
(tests/test_gpu_reference_sources.py::test_reference_script_replay) replays the recorded invocations one by one against the
binaries built from the reference's own sources. The trace is data derived by running the scripts -- environment settings
and program names -- not their text.

    python trace_scripts.py [REF=/root/reference] [OUT=_build/script_trace.json]
"""
import json
import os
import stat
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
STUB = """#!/bin/bash
python3 - "$0" "$@" <<'EOF'
import json, os, sys
base = json.load(open(os.environ["HEXL_TRACE_BASE_ENV"]))
env = {k: v for k, v in os.environ.items() if base.get(k) != v and not k.startswith("HEXL_TRACE_") and k not in ("_", "SHLVL", "PWD", "OLDPWD")}
with open(os.environ["HEXL_TRACE_FILE"], "a") as f:
    f.write(json.dumps({"script": os.environ["HEXL_TRACE_SCRIPT"], "exe": os.path.basename(sys.argv[1]), "argv": sys.argv[2:], "env": env}) + "\\n")
EOF
"""


def trace(ref: Path, out: Path) -> int:
    scripts = sorted(ref.glob("tests/micro_*.sh")) + sorted(ref.glob("benchmark/micro_*.sh"))
    if not scripts:
        print(f"no runner scripts under {ref}", file=sys.stderr)
        return 1
    entries = []
    with tempfile.TemporaryDirectory(prefix="hexl_trace_") as tmp:
        tmp = Path(tmp)
        for name in ("test_fwd_ntt", "test_inv_ntt", "test_dyadic_multiply", "test_keyswitch", "test_dyadic_multiply_keyswitch",
                     "bench_fwd_ntt", "bench_inv_ntt", "bench_dyadic_multiply", "bench_keyswitch"):
            p = tmp / name
            p.write_text(STUB)
            p.chmod(p.stat().st_mode | stat.S_IEXEC)
        # what the scripts are started with; a stub records only what the script itself set on top of this
        env = {k: v for k, v in os.environ.items() if not k.startswith(("FPGA_", "BATCH_SIZE_")) and k not in ("N", "ITER", "NUM_DEV")}
        env.update(RUN_CHOICE="1", FPGA_BITSTREAM_DIR="/nonexistent-bitstreams")
        base = tmp / "base_env.json"
        base.write_text(json.dumps(env))
        for s in scripts:
            tf = tmp / "trace.jsonl"
            if tf.exists():
                tf.unlink()
            rel = str(s.relative_to(ref))
            r = subprocess.run(["bash", str(s)], cwd=tmp, capture_output=True, text=True, timeout=120,
                               env=dict(env, HEXL_TRACE_FILE=str(tf), HEXL_TRACE_BASE_ENV=str(base), HEXL_TRACE_SCRIPT=rel))
            if r.returncode != 0:
                print(f"{rel} exited {r.returncode}: {r.stderr[-400:]}", file=sys.stderr)
                return 1
            got = [json.loads(l) for l in tf.read_text().splitlines()] if tf.exists() else []
            if not got:
                print(f"{rel} called none of the known binaries", file=sys.stderr)
                return 1
            entries += got
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps({"run_with": {"RUN_CHOICE": "1", "FPGA_BITSTREAM_DIR": "/nonexistent-bitstreams"},
                               "scripts": sorted({e["script"] for e in entries}), "invocations": entries}, indent=1))
    print(f"traced {len(scripts)} scripts, {len(entries)} invocations -> {out}")
    return 0


if __name__ == "__main__":
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    out = Path(sys.argv[2] if len(sys.argv) > 2 else HERE / "_build" / "script_trace.json")
    sys.exit(trace(ref, out))
