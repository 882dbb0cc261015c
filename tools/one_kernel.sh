#!/bin/bash
# one_kernel.sh [hipcc flags] -- compile ONLY k_ksx_main<14,4,3,false,true> (the BASELINE workload's kernel) from keyswitch_x.hip:
# resource usage, scratch operations with their line numbers in /tmp/one.s (a 15 s turn-around instead of two minutes for the file)
cd "$(dirname "$0")/../hexl-fpga_amd/csrc"
END=$(grep -n "^template <class K>" keyswitch_x.hip | head -1 | cut -d: -f1)
head -n $((END-2)) keyswitch_x.hip > /tmp/kx_one.hip
echo "template __global__ void k_ksx_main<14, 4, 3, false, true${DL:+, true}>(KsArgsX);" >> /tmp/kx_one.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -I../../include --cuda-device-only -S -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/one.s /tmp/kx_one.hip 2>&1 |
  grep -E "VGPRs:|VGPRs Spill|ScratchSize|SGPRs Spill|error" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - -
grep -n "scratch_\|Loop Header\|s_barrier" /tmp/one.s | head -80
