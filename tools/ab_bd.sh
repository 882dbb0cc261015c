#!/bin/bash
# ab_bd.sh <out> <variant...> -- the (b, d)-major FP64 pipeline: batch 32 at N = 16384, N = 32768 at L = 3, and HEXL_KS_PIPE=1 at batch 2048
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for v in shipped "$@"; do
    if [ $v = shipped ]; then unset HEXL_MI355X_LIB; else export HEXL_MI355X_LIB=$R/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so; fi
    echo -n "$v bd-major b32: " >> $OUT; python $R/tools/ks_rate.py 32 7 51 200 2>&1 | grep parity >> $OUT
    echo -n "$v n32768: " >> $OUT; python $R/tools/ks_rate.py 2048 3 51 10 32768 2>&1 | grep parity >> $OUT
    echo -n "$v pipe1 b2048: " >> $OUT; HEXL_KS_PIPE=1 python $R/tools/ks_rate.py 2048 7 51 10 2>&1 | grep parity >> $OUT
  done
done
cat $OUT
