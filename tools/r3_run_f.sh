#!/bin/bash
# round 3, GPU session F: persistent k_ksx_main_p (out-of-line item call)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
HEXL_KSX_MAIN_PERSIST=1 python -m pytest tests/test_gpu_keyswitch.py tests/test_gpu_mulrelin.py -x -q > $O/pytest_persist.log 2>&1; tail -2 $O/pytest_persist.log
for rep in 1 2; do
python tools/ks_rate.py 4096 7 51 10 > $O/rate_p0_$rep.txt 2>&1; echo "persist0: $(tail -1 $O/rate_p0_$rep.txt)"
HEXL_KSX_MAIN_PERSIST=1 python tools/ks_rate.py 4096 7 51 10 > $O/rate_p1_$rep.txt 2>&1; echo "persist1: $(tail -1 $O/rate_p1_$rep.txt)"
done
HEXL_KSX_MAIN_PERSIST=1 python tools/ks_rate.py 1024 7 51 20 > $O/rate_p1_b1024.txt 2>&1; echo "persist1 b1024: $(tail -1 $O/rate_p1_b1024.txt)"
python tools/ks_rate.py 1024 7 51 20 > $O/rate_p0_b1024.txt 2>&1; echo "persist0 b1024: $(tail -1 $O/rate_p0_b1024.txt)"
HEXL_KSX_MAIN_PERSIST=1 python tools/ks_rate.py 4096 6 51 10 > $O/rate_p1_L6.txt 2>&1; echo "persist1 L6: $(tail -1 $O/rate_p1_L6.txt)"
python tools/ks_rate.py 4096 6 51 10 > $O/rate_p0_L6.txt 2>&1; echo "persist0 L6: $(tail -1 $O/rate_p0_L6.txt)"
