#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
for rep in 1 2; do
for t in 2 4 8 12; do for mb in 32 64 128; do for ws in 256 1024; do
echo "rep=$rep threads=$t sub_mb=$mb ws=$ws: $(HEXL_HOST_THREADS=$t HEXL_HOST_SUB_MB=$mb tests/cpp/bench_cxx_api $ws 6 2>&1 | grep keyswitch | sed 's/.*batch=[0-9]*: //')" | tee -a $O/host_sweep.txt
done; done; done; done
