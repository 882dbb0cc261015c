mkdir -p gpurun_out/r4c
for rep in 1 2; do for v in base asmrd; do
  if [ $v = base ]; then unset HEXL_MI355X_LIB; else export HEXL_MI355X_LIB=$PWD/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so; fi
  echo "== $v rep $rep: $(timeout 300 python tools/ks_rate.py 8192 7 51 10 2>&1 | tail -1)"
done; done > gpurun_out/r4c/variants.txt 2>&1
unset HEXL_MI355X_LIB
cat gpurun_out/r4c/variants.txt
for L in 6 7; do
  echo "--- L=$L latency path (default)"; timeout 300 python tools/batch_sweep.py $L 1,2,3,4,6,8,12,16 2>&1 | grep batch
  echo "--- L=$L five kernels (HEXL_KS_LAT=0)"; HEXL_KS_LAT=0 timeout 300 python tools/batch_sweep.py $L 1,2,3,4,6,8,12,16 2>&1 | grep batch
done > gpurun_out/r4c/latency.txt 2>&1
cat gpurun_out/r4c/latency.txt
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c/pytest.log); tail -6 gpurun_out/r4c/pytest.log
(HEXL_KS_LAT=1 timeout 900 python -m pytest tests/test_gpu_keyswitch.py -m gpu -q -x -k "vs_oracle or arithmetic_path or fused_and_per or caller_twiddles or range_flag" > gpurun_out/r4c/pytest_lat1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c/pytest_lat1.log); tail -4 gpurun_out/r4c/pytest_lat1.log
timeout 600 python tools/byte_budget.py --out gpurun_out/r4c/bytes --masks 0,1,2,8,31 > gpurun_out/r4c/bytes.log 2>&1; tail -7 gpurun_out/r4c/bytes.log
timeout 600 python bench.py --steps 10 --no-pmc --no-extra > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; tail -c 300 gpurun_out/r4c/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c/bench.json"))
c=d["cpu_baseline"]
print(d["value"], c["by_threads"], c["parallel_efficiency"], c["host_stream_triad_GBps_by_threads"], c["numa_nodes_used"])
PY
