#!/bin/bash
# pmc_int.sh <tag> -- VALU wave-instructions per keyswitch and issue fraction of the INTEGER slot-major kernels (k_ksi_*; moduli in [2^52, 2^60),
# here forced with HEXL_KS_INT=1 on the benchmark's 51-bit primes: the instruction stream does not depend on the modulus) in both
# geometries, 32 x 512 (default) and 16 x 1024 (HEXL_KSI_LOGE=4). Same workload and counter groups as tools/pmc_tier.sh. VERDICT r05 item 8.
TAG=${1:-pmc_int}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rm -f $OUT/summary.txt
for loge in 5 4; do
  rm -rf $OUT/e$loge
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    HEXL_KS_INT=1 HEXL_KSI_LOGE=$loge HEXL_KS_ONE_LANE=1 rocprofv3 --kernel-trace --pmc $set -d $OUT/e$loge/p$i -- $R/tools/pmc_workload 256 7 2 > $OUT/e${loge}_p$i.log 2>&1
  done
  echo "==== integer kernels, 2^$loge coefficients per thread" >> $OUT/summary.txt
  python3 $R/tools/pmc_summary.py $OUT/e$loge 256 7 >> $OUT/summary.txt 2>&1
  rm -rf $OUT/e$loge
done
cat $OUT/summary.txt
