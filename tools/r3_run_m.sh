#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
python tools/ks_rate.py 8192 7 51 6 16384 2>&1 | tail -1 | tee -a $O/n_at_L7.txt
python tools/ks_rate.py 16384 7 51 6 8192 2>&1 | tail -1 | tee -a $O/n_at_L7.txt
python tools/ks_rate.py 32768 7 51 6 4096 2>&1 | tail -1 | tee -a $O/n_at_L7.txt
python tools/ks_rate.py 16384 6 51 6 8192 2>&1 | tail -1 | tee -a $O/n_at_L7.txt
