mkdir -p gpurun_out/r4f; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4f/kt -- python $R/tools/batch_sweep.py 6 1 > $R/gpurun_out/r4f/trace.log 2>&1
python3 $R/tools/rocprof_summary.py $(find $R/gpurun_out/r4f/kt -name "*.db" | head -1) 2>&1 | head -7 | cut -c1-150
rm -rf $R/gpurun_out/r4f/kt
cd $R
python tools/lat_probe.py 6 2
python tools/lat_probe.py 7 2
HEXL_KS_LAT=0 python tools/lat_probe.py 6 2
(timeout 900 python -m pytest tests/test_gpu_keyswitch.py -m gpu -q -x -k "latency" > gpurun_out/r4f/pytest_lat.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4f/pytest_lat.log); tail -3 gpurun_out/r4f/pytest_lat.log
