mkdir -p gpurun_out/r4b
for rep in 1 2; do for v in base lds1 lds3 pref lds1pref; do
  if [ $v = base ]; then unset HEXL_MI355X_LIB; else export HEXL_MI355X_LIB=$PWD/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so; fi
  echo "== $v rep $rep: $(timeout 300 python tools/ks_rate.py 8192 7 51 10 2>&1 | tail -1)"
done; done > gpurun_out/r4b/variants.txt 2>&1
unset HEXL_MI355X_LIB
cat gpurun_out/r4b/variants.txt
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4b/pytest.log); tail -8 gpurun_out/r4b/pytest.log
timeout 900 python tools/byte_budget.py --out gpurun_out/r4b/bytes > gpurun_out/r4b/bytes.log 2>&1; tail -9 gpurun_out/r4b/bytes.log
timeout 600 python bench.py --steps 10 --no-pmc > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err; tail -c 400 gpurun_out/r4b/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4b/bench.json"))
print(d["value"], d["cpu_baseline"]["by_threads"], d["cpu_baseline"]["parallel_efficiency"], d["cpu_baseline"]["numa_nodes_used"])
for k,v in d["extra"].items():
    if k.startswith("ntt_") or k.startswith("cxx"): print(k, json.dumps(v)[:400])
PY
