#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=${1:-$R/gpurun_out/fp64power}; mkdir -p $O; cd $R
for mix in 0 1; do
  tools/fp64_power 12 $mix > $O/burn$mix.txt 2>&1 &
  PID=$!
  sleep 5
  for i in $(seq 1 8); do echo "--- sample $i" >> $O/load_burn$mix.txt; rocm-smi --showpower --showclocks --showtemp >> $O/load_burn$mix.txt 2>&1; sleep 0.5; done
  wait $PID
  echo "== mix $mix: $(tail -1 $O/burn$mix.txt)"
  grep -E "Package Power|sclk" $O/load_burn$mix.txt | sed 's/.*: //' | sort | uniq -c | sort -rn | head -5
done
