mkdir -p gpurun_out/r4g
for L in 6 7; do
  echo "--- L=$L HEXL_KS_LAT=2 (quarter transforms, batched)"; HEXL_KS_LAT=2 timeout 300 python tools/batch_sweep.py $L 1,2,3,4,6,8,12,16,24,32 2>&1 | grep batch
  echo "--- L=$L HEXL_KS_LAT=0 (five kernels / slot-major)"; HEXL_KS_LAT=0 timeout 300 python tools/batch_sweep.py $L 1,2,3,4,6,8,12,16,24,32 2>&1 | grep batch
done > gpurun_out/r4g/latency.txt 2>&1
cat gpurun_out/r4g/latency.txt
(timeout 900 python -m pytest tests/test_gpu_keyswitch.py -m gpu -q -x -k "latency" > gpurun_out/r4g/pytest_lat.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4g/pytest_lat.log); tail -3 gpurun_out/r4g/pytest_lat.log
