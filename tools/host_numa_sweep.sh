#!/bin/bash
# host_numa_sweep.sh -- host-pointer KeySwitch at worksize 128 (tests/cpp/bench_cxx_api) under copy-thread counts and NUMA placements:
# is the host's accumulate (the slowest stage) limited by threads, by the pod's CPU quota or by cross-socket traffic?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=$R/tests/cpp/bench_cxx_api
echo "# nodes: $(ls -d /sys/devices/system/node/node* | wc -l); quota: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for n in /sys/devices/system/node/node*; do echo "# $(basename $n): cpus $(cat $n/cpulist)"; done
echo "# GPU 0 numa node: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
for round in 1 2; do
  for th in 8 12 16; do
    echo "free   threads $th: $(HEXL_HOST_THREADS=$th timeout 120 $B 128 6 0 1 | tail -1)"
  done
  for n in /sys/devices/system/node/node*; do
    for th in 8 16; do
      echo "$(basename $n) threads $th: $(HEXL_HOST_THREADS=$th timeout 120 taskset -c $(cat $n/cpulist) $B 128 6 0 1 | tail -1)"
    done
  done
done
