#!/bin/bash
# profile_round.sh <tag> -- the round's evidence, written under gpurun_out/<tag>/ on the GPU box:
#   kernel trace of the default bench.py run, PMC passes (one counter group per run, with --kernel-trace only)
#   of tools/profile_ks.py, the text summaries and the traffic JSON that bench.py reads.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/kt $OUT/pmc
# one lane: the default schedule alternates chunks between two streams, whose kernels then overlap and stretch each
# other's durations in the trace; one lane gives per-kernel durations that add up (bench.py's stage timers are one-lane too)
HEXL_KS_ONE_LANE=1 rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --no-cpu --no-pmc > $OUT/bench_under_trace.log 2>&1
# bench.py starts child processes (tests/cpp/bench_cxx_api for the end-to-end leg), each with its own database: the largest one is bench.py's
DB=$(find $OUT/kt -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2-)
python3 $R/tools/rocprof_summary.py $DB > $OUT/kernel_trace.txt 2>&1
python3 $R/tools/overlap.py $DB >> $OUT/kernel_trace.txt 2>&1
for other in $(find $OUT/kt -name "*.db" | grep -v "$DB"); do
  echo "" >> $OUT/kernel_trace.txt; echo "# child process $(basename $other) (tests/cpp/bench_cxx_api: the lone-keyswitch latency path, keyswitch_lat.hip)" >> $OUT/kernel_trace.txt
  python3 $R/tools/rocprof_summary.py $other 2>&1 | head -8 >> $OUT/kernel_trace.txt
done
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  # one lane: every dispatch is one whole chunk of 256 keyswitches, so per-dispatch averages divide by 256
  HEXL_KS_ONE_LANE=1 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc/p$i -- python $R/tools/profile_ks.py 256 7 > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $OUT/pmc 256 7 > $OUT/pmc.txt 2>&1
cp $OUT/pmc/traffic.json $OUT/traffic.json 2>/dev/null
cp $OUT/pmc/alu.json $OUT/alu.json 2>/dev/null
rm -rf $OUT/pmc $OUT/kt
python $R/bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
tail -3 $OUT/pmc.txt
