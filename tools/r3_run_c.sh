#!/bin/bash
# round 3, GPU session C: next input through LDS (k_ksx_main<..., DMA>) -- parity, rate
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3c; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_keyswitch.py -x -q > $O/pytest_default.log 2>&1; tail -2 $O/pytest_default.log
for rep in 1 2; do
HEXL_KSX_DMA=0 python tools/ks_rate.py 4096 7 51 10 > $O/rate_dma0_$rep.txt 2>&1; echo "dma0: $(tail -1 $O/rate_dma0_$rep.txt)"
python tools/ks_rate.py 4096 7 51 10 > $O/rate_dma1_$rep.txt 2>&1; echo "dma1: $(tail -1 $O/rate_dma1_$rep.txt)"
done
for v in pf2 pf4; do
  HEXL_MI355X_LIB=$R/hexl-fpga_amd/lib_var/$v/libhexl_mi355x.so python tools/ks_rate.py 4096 7 51 10 > $O/rate_$v.txt 2>&1; echo "$v: $(tail -1 $O/rate_$v.txt)"
done
python tools/ks_rate.py 4096 6 51 10 > $O/rate_L6.txt 2>&1; tail -1 $O/rate_L6.txt
python tools/ks_rate.py 4096 6 48 10 > $O/rate_L6_48.txt 2>&1; tail -1 $O/rate_L6_48.txt
python tools/ks_rate.py 4096 7 52 10 > $O/rate_L7_strict.txt 2>&1; tail -1 $O/rate_L7_strict.txt
