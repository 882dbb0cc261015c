#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
for t in 8 16 32 64 128; do for ws in 256 1024; do
echo "threads=$t ws=$ws: $(HEXL_HOST_THREADS=$t tests/cpp/bench_cxx_api $ws 6 2>&1 | grep keyswitch)" | tee -a $O/host_threads.txt
done; done
for mb in 8 16 64; do echo "sub_mb=$mb ws=1024: $(HEXL_HOST_SUB_MB=$mb tests/cpp/bench_cxx_api 1024 6 2>&1 | grep keyswitch)" | tee -a $O/host_threads.txt; done
