#!/bin/bash
# power_probe.sh [out] -- board power, clocks and temperature (rocm-smi, every 0.5 s) while one kernel family runs in a loop:
# is the shader clock under the FP64 load a power-management outcome? (It is: DESIGN.md 4.5.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=${1:-$R/gpurun_out/power}; mkdir -p $O; cd $R
rocm-smi --showpower --showclocks --showtemp > $O/idle.txt 2>&1
rocm-smi --showmaxpower > $O/caps.txt 2>&1
for kind in ks ntt inv dyadic; do
  python tools/load_loop.py $kind 14 > $O/$kind.txt 2>&1 &
  PID=$!
  while ! grep -q START $O/$kind.txt 2>/dev/null; do sleep 0.5; kill -0 $PID 2>/dev/null || break; done
  sleep 4
  for i in $(seq 1 10); do echo "--- sample $i" >> $O/load_$kind.txt; rocm-smi --showpower --showclocks --showtemp >> $O/load_$kind.txt 2>&1; sleep 0.5; done
  wait $PID
  echo "== $kind: $(tail -1 $O/$kind.txt)"
  grep -E "Package Power|sclk" $O/load_$kind.txt | sed 's/.*: //' | sort | uniq -c | sort -rn | head -6
done
grep "Max" $O/caps.txt
