#!/bin/bash
# kres.sh <file.hip> [name filter] [extra hipcc flags] -- registers, scratch and spills per kernel of one translation unit
# (-Rpass-analysis=kernel-resource-usage), one line per kernel
F=${1:-keyswitch_x.hip}; PAT=${2:-.}; shift 2
cd "$(dirname "$0")/../hexl-fpga_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c $F -o /tmp/kres_$$.o 2>&1 |
  grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|SGPRs Spill" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - | grep -E "$PAT" | c++filt | sed -E 's/Function Name: void //'
rm -f /tmp/kres_$$.o
