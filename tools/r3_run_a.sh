#!/bin/bash
# first GPU session of round 3: test suite, default bench (in-run PMC), config-5 pre-flight on one GPU, host API, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3a; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
for B in 1024 2048 8192; do
  python bench.py --total-batch $B --steps 50 --no-extra --no-cpu --no-pmc > $O/b$B.json 2>> $O/bench.err
  python bench.py --total-batch $B --steps 50 --barrier-per-step --no-extra --no-cpu --no-pmc > $O/b${B}_barrier.json 2>> $O/bench.err
done
HEXL_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --total-batch 2048 --steps 50 --no-cpu --no-pmc > $O/two_ranks_one_gpu.json 2>> $O/bench.err
for ws in 1 2 8 32 256; do tests/cpp/bench_cxx_api $ws 6 >> $O/cxx_api.txt 2>&1; done
NUM_DEV=2 HEXL_DEV_ALIAS=1 tests/cpp/bench_cxx_api 256 6 >> $O/cxx_api.txt 2>&1
tools/ksx_timeline 256 7 > $O/timeline.txt 2>&1
tail -3 $O/pytest.log; tail -c 600 $O/bench.json; cat $O/cxx_api.txt | grep keyswitch
