#!/bin/bash
# pmc_quick.sh <tag> [batch] [L] -- PMC passes (one counter group per run, --kernel-trace only) of tools/profile_ks.py
# into gpurun_out/<tag>/, summarised by tools/pmc_summary.py. Extra environment (HEXL_KS_PIPE=...) is inherited.
TAG=${1:-pmcq}; B=${2:-256}; L=${3:-7}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  HEXL_KS_ONE_LANE=1 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc/p$i -- python $R/tools/profile_ks.py $B $L > $OUT/pmc_p$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT/pmc $B $L > $OUT/pmc.txt 2>&1
cp $OUT/pmc/traffic.json $OUT/traffic.json 2>/dev/null
cp $OUT/pmc/alu.json $OUT/alu.json 2>/dev/null
rm -rf $OUT/pmc
tail -5 $OUT/pmc.txt
