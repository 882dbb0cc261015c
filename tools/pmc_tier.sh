#!/bin/bash
# pmc_tier.sh <tag> -- VALU wave-instructions per keyswitch and FP64-issue fraction of the slot-major pipeline for the LAZY tier (smallest
# 52-bit primes: bench.py's headline) and the STRICT tier (largest 52-bit primes, HEXL_WORKLOAD_PRIMES=top52), same box, same workload
# (tools/pmc_workload 256 7 2), one counter group per rocprofv3 run (--kernel-trace only). VERDICT r05 item 4.
TAG=${1:-pmc_tier}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
for tier in lazy strict; do
  rm -rf $OUT/$tier
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    if [ $tier = strict ]; then export HEXL_WORKLOAD_PRIMES=top52; else unset HEXL_WORKLOAD_PRIMES; fi
    HEXL_KS_ONE_LANE=1 rocprofv3 --kernel-trace --pmc $set -d $OUT/$tier/p$i -- $R/tools/pmc_workload 256 7 2 > $OUT/${tier}_p$i.log 2>&1
  done
  echo "==== $tier tier" >> $OUT/summary.txt
  python3 $R/tools/pmc_summary.py $OUT/$tier 256 7 >> $OUT/summary.txt 2>&1
  rm -rf $OUT/$tier
done
cat $OUT/summary.txt
