#!/bin/bash
# ab_env.sh <out> <VAR=value ...> -- tools/ks_rate.py (batch 8192, L = 7) for the shipped library as it is and under each environment
# setting, two interleaved rounds on one box
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for v in default "$@"; do
    echo -n "$v: " >> $OUT
    if [ $v = default ]; then python $R/tools/ks_rate.py 8192 7 51 40 2>&1 | grep parity >> $OUT
    else env $v python $R/tools/ks_rate.py 8192 7 51 40 2>&1 | grep parity >> $OUT; fi
  done
done
cat $OUT
