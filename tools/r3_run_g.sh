#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R
for m in 0 1 2; do
HEXL_KSX_MAIN_PERSIST=$m python tools/ks_rate.py 4096 7 51 10 > $O/rate_p$m.txt 2>&1; echo "persist=$m: $(tail -1 $O/rate_p$m.txt)"
HEXL_KS_ONE_LANE=1 HEXL_KSX_MAIN_PERSIST=$m python tools/ks_rate.py 4096 7 51 10 > $O/rate_p${m}_onelane.txt 2>&1; echo "persist=$m one lane: $(tail -1 $O/rate_p${m}_onelane.txt)"
done
cd /tmp; export TMPDIR=/tmp
HEXL_KS_ONE_LANE=1 HEXL_KSX_MAIN_PERSIST=1 rocprofv3 --kernel-trace --stats -d $O/kt -- $R/tools/pmc_workload 256 7 3 > $O/kt.log 2>&1
python3 $R/tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_trace_persist1.txt 2>&1; rm -rf $O/kt; head -6 $O/kernel_trace_persist1.txt | cut -c1-150
