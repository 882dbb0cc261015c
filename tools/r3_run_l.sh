#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3l; mkdir -p $O; cd $R
for rep in 1 2; do for ws in 1 8 32 256 1024; do echo "new defaults ws=$ws: $(tests/cpp/bench_cxx_api $ws 6 2>&1 | grep "end-to-end" | sed 's/C++ API end-to-end //' | tr '\n' '|')" | tee -a $O/host_new.txt; done; done
for ws in 256 1024; do echo "old defaults ws=$ws: $(HEXL_HOST_THREADS=32 HEXL_HOST_SUB_MB=32 tests/cpp/bench_cxx_api $ws 6 2>&1 | grep "end-to-end" | sed 's/C++ API end-to-end //' | tr '\n' '|')" | tee -a $O/host_new.txt; done
python -m pytest tests/test_gpu_cxx_api.py tests/test_gpu_host_api.py tests/test_gpu_reference_sources.py -x -q 2>&1 | tail -2
