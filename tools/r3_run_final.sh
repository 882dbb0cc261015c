#!/bin/bash
# the round's closing check on a fresh box: GPU suite, smoke, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3final; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
