#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_cxx_api.py tests/test_gpu_host_api.py tests/test_gpu_reference_sources.py tests/test_cmake_package.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do for ws in 1 2 8 32; do echo "ws=$ws: $(tests/cpp/bench_cxx_api $ws 6 2>&1 | grep "keyswitch N" | sed 's/C++ API end-to-end //')" | tee -a $O/host_small.txt; done; done
HEXL_HOST_TRACE=1 tests/cpp/bench_cxx_api 1 6 2>&1 | grep "hexl host" | sed -n 5,12p
