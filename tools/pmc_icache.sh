#!/bin/bash
# pmc_icache.sh <tag> -- instruction-cache counters of the keyswitch kernels (tools/pmc_workload, chunk of 256, L = 7), one group per run
TAG=${1:-ic}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_INPUT_VALID_READYB" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM"; do
  i=$((i+1)); rm -rf $OUT/p$i
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o r -- $R/tools/pmc_workload 256 7 2 0 > $OUT/p$i.log 2>&1
done
python3 - $OUT $R > $OUT/icache.txt <<'PY'
import sys
sys.path.insert(0, sys.argv[2] + '/tools')
import pmc_summary
vals, dur = pmc_summary.collect(sys.argv[1])
for k in sorted(vals):
    print(k, ' avg us under PMC:', ' '.join('%.1f' % d for d in dur.get(k, [])))
    for c in sorted(vals[k]):
        print(f"    {c:34s} {vals[k][c]:14.5g}")
PY
cat $OUT/icache.txt
rm -rf $OUT/p?
