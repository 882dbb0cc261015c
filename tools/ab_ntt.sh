#!/bin/bash
# ab_ntt.sh <out> <variant...> -- standalone NTT rates (bench.time_ntt: N = 16384, batch 1024 and 4096, 300 launches) for the shipped
# library and the named variant libraries, two interleaved rounds on one box
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
for b in (1024, 4096):
    r = bench.time_ntt(hx, ctx, orc, dev, b, 300)
    print('$v', b, 'fwd %.2f M/s  inv %.2f M/s' % (r['fwd']['ntt_per_s'] / 1e6, r['inv']['ntt_per_s'] / 1e6))
PY
  done
done
cat $OUT
