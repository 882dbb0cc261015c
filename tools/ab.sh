#!/bin/bash
# ab.sh <out> <variant...> -- tools/ks_rate.py (batch 8192, L = 7, 51-bit primes, 40 repetitions) for the shipped library and the
# named variant libraries (hexl-fpga_amd/lib_var/<name>), two interleaved rounds on one box
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for v in shipped "$@"; do
    if [ $v = shipped ]; then unset HEXL_MI355X_LIB; else export HEXL_MI355X_LIB=$R/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so; fi
    echo -n "$v: " >> $OUT
    python $R/tools/ks_rate.py 8192 7 51 40 2>&1 | grep parity >> $OUT
  done
done
cat $OUT
