mkdir -p gpurun_out/r4d
(timeout 900 python -m pytest tests/test_gpu_keyswitch.py -m gpu -q -x -k "latency or vs_oracle or rlwe or range_flag or batch_chunks" > gpurun_out/r4d/pytest_lat.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4d/pytest_lat.log); tail -15 gpurun_out/r4d/pytest_lat.log
for L in 6 7; do
  echo "--- L=$L default (quarter transforms at batch 1)"; timeout 300 python tools/batch_sweep.py $L 1,2,4 2>&1 | grep batch
  echo "--- L=$L HEXL_KS_LAT=2 (quarter transforms, instance by instance)"; HEXL_KS_LAT=2 timeout 300 python tools/batch_sweep.py $L 1,2,4,8 2>&1 | grep batch
  echo "--- L=$L HEXL_KS_LAT=1 (three kernels)"; HEXL_KS_LAT=1 timeout 300 python tools/batch_sweep.py $L 1,2 2>&1 | grep batch
  echo "--- L=$L HEXL_KS_LAT=0 (five kernels)"; HEXL_KS_LAT=0 timeout 300 python tools/batch_sweep.py $L 1,2,4,8 2>&1 | grep batch
done > gpurun_out/r4d/latency.txt 2>&1
cat gpurun_out/r4d/latency.txt
for rep in 1 2; do for v in base pref2 pref4 persist prio4 prio8 prio1; do
  if [ $v = base ]; then unset HEXL_MI355X_LIB; else export HEXL_MI355X_LIB=$PWD/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so; fi
  echo "== $v rep $rep: $(timeout 300 python tools/ks_rate.py 8192 7 51 10 2>&1 | tail -1)"
done; done > gpurun_out/r4d/variants.txt 2>&1
unset HEXL_MI355X_LIB
cat gpurun_out/r4d/variants.txt
timeout 600 python tools/byte_budget.py --out gpurun_out/r4d/bytes --masks 0,1,2,4,8,16,31 > gpurun_out/r4d/bytes.log 2>&1; tail -9 gpurun_out/r4d/bytes.log
cd tests/cpp && for ws in 1 2 16; do ./bench_cxx_api $ws 6 0 1 2>/dev/null | tail -1; done
