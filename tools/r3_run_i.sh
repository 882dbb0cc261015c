#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_ntt.py tests/test_gpu_host_api.py tests/test_gpu_cxx_api.py tests/test_gpu_reference_sources.py -x -q > $O/pytest_ntt.log 2>&1; tail -3 $O/pytest_ntt.log
for rep in 1 2; do
HEXL_NTT_TABLE_CACHE=0 python tools/ntt_rate.py > $O/ntt_cache0_$rep.txt 2>&1; echo "cache0: $(grep -o "'ntt_per_s': [0-9.]*" $O/ntt_cache0_$rep.txt | tr '\n' ' ')"
python tools/ntt_rate.py > $O/ntt_cache1_$rep.txt 2>&1; echo "cache1: $(grep -o "'ntt_per_s': [0-9.]*" $O/ntt_cache1_$rep.txt | tr '\n' ' ')"
done
