mkdir -p gpurun_out/r4h
(timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4h/pytest.log); tail -5 gpurun_out/r4h/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r4h_prof > gpurun_out/r4h/profile_round.log 2>&1; tail -5 gpurun_out/r4h/profile_round.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4h_prof/bench.json").read())
print("value", d["value"], "alu", d["roofline"]["alu"]["achieved_frac"], d["roofline"]["alu"].get("achieved_frac_at_pmc_pass_clock"), "mJ", d["roofline"].get("energy_mj_per_keyswitch"), "traffic/ks", d["roofline"]["traffic"]/8192/1e6)
print("ntt", d.get("ntt_fwd_per_s"), d.get("ntt_inv_per_s"))
c=d["cpu_baseline"]; print("cpu", c["value"], c["cores"], c["by_threads"], c["host_stream_triad_GBps_by_threads"], c["cgroup_cpu_quota_cores"])
print("e2e", json.dumps(d["extra"]["cxx_api_end_to_end"])[:600])
PY
cd tests/cpp && for ws in 1 2 3 16; do ./bench_cxx_api $ws 6 0 1 2>/dev/null | tail -1; done
