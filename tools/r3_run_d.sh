#!/bin/bash
# round 3, GPU session D: full GPU suite after the range flag / mulrelin / cmake changes; host trace at worksize 1
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
HEXL_HOST_TRACE=1 tests/cpp/bench_cxx_api 1 6 > $O/trace_ws1.txt 2>&1; grep -c "hexl host" $O/trace_ws1.txt; tail -40 $O/trace_ws1.txt | head -30
