#!/bin/bash
# round 3, GPU session B: k_ksx_main2 variants -- parity, rate, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_keyswitch.py -x -q > $O/pytest_default.log 2>&1; tail -2 $O/pytest_default.log
HEXL_KSX_MAIN=1 python tools/ks_rate.py 4096 7 51 10 > $O/rate_main1.txt 2>&1; tail -1 $O/rate_main1.txt
python tools/ks_rate.py 4096 7 51 10 > $O/rate_default.txt 2>&1; tail -1 $O/rate_default.txt
for v in b_pre11 c_pre0 d_pre10 e_pf3xq2 f_xq2 g_pre1; do
  HEXL_MI355X_LIB=$R/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so python tools/ks_rate.py 4096 7 51 10 > $O/rate_$v.txt 2>&1; echo "$v: $(tail -1 $O/rate_$v.txt)"
done
HEXL_KSX_MAIN=1 python tools/ks_rate.py 4096 7 51 10 > $O/rate_main1_again.txt 2>&1; tail -1 $O/rate_main1_again.txt
tools/ksx_timeline 256 7 > $O/timeline.txt 2>&1
python tools/ks_rate.py 4096 6 51 10 > $O/rate_default_L6.txt 2>&1; tail -1 $O/rate_default_L6.txt
python tools/ks_rate.py 4096 6 48 10 > $O/rate_default_L6_48.txt 2>&1; tail -1 $O/rate_default_L6_48.txt
