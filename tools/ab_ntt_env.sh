#!/bin/bash
# ab_ntt_env.sh <out> <VAR=value> -- standalone NTT rates (bench.time_ntt: N = 16384, batch 1024 and 4096, 300 launches) with and without
# one environment knob, three interleaved rounds on one box (e.g. HEXL_NTT_PERSIST=0: one workgroup per polynomial)
OUT=$1; KNOB=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $(dirname $OUT); : > $OUT
for round in 1 2 3; do
  for v in default "$KNOB"; do
    ( if [ "$v" != default ]; then export "$v"; fi
    python - >> $OUT 2>/dev/null <<PY
import sys
sys.path[:0]=['$R','$R/oracle','$R/tests']
import torch, hexl_fpga_amd as hx, orc, bench
dev=torch.device('cuda:0'); ctx=hx.Context(0)
for b in (1024, 4096):
    r = bench.time_ntt(hx, ctx, orc, dev, b, 300)
    print('%-28s' % '$v', b, 'fwd %.2f M/s  inv %.2f M/s' % (r['fwd']['ntt_per_s'] / 1e6, r['inv']['ntt_per_s'] / 1e6))
PY
    )
  done
done
cat $OUT
