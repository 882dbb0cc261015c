#!/bin/bash
# ab_misc.sh <out> <variant...> -- the secondary paths for the shipped library and variant libraries: integer-kernel NTT (q = 2^52 + 393217,
# batch 1024), integer keyswitch (HEXL_KS_INT=1), (b, d)-major FP64 keyswitch (batch 32; N = 32768 at L = 3)
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2; do
  for v in shipped "$@"; do
    if [ $v = shipped ]; then unset HEXL_MI355X_LIB; else export HEXL_MI355X_LIB=$R/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so; fi
    python - >> $OUT 2>/dev/null <<PY
import sys
sys.path[:0]=['$R','$R/oracle','$R/tests']
import torch, hexl_fpga_amd as hx, orc, bench
dev=torch.device('cuda:0'); ctx=hx.Context(0)
r = bench.time_ntt(hx, ctx, orc, dev, 1024, 300, q=4503599627763713)
print('$v', 'integer NTT batch 1024: fwd %.2f M/s  inv %.2f M/s' % (r['fwd']['ntt_per_s'] / 1e6, r['inv']['ntt_per_s'] / 1e6))
PY
    echo -n "$v int-ks: " >> $OUT; HEXL_KS_INT=1 python $R/tools/ks_rate.py 2048 7 51 10 2>&1 | grep parity >> $OUT
    echo -n "$v bd-major b32: " >> $OUT; python $R/tools/ks_rate.py 32 7 51 200 2>&1 | grep parity >> $OUT
    echo -n "$v n32768: " >> $OUT; python $R/tools/ks_rate.py 2048 3 51 10 32768 2>&1 | grep parity >> $OUT
  done
done
cat $OUT
