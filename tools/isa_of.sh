#!/bin/bash
# isa_of.sh <file.hip> <mangled-name regex> [extra flags] -- gfx950 ISA of one kernel into /tmp/isa.s, with block structure and spill sites
F=$1; PAT=$2; shift 2
cd "$(dirname "$0")/../hexl-fpga_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S "$@" -o /tmp/all.s $F 2>/dev/null
NAME=$(grep -E "^_Z.*:" /tmp/all.s | grep -E "$PAT" | head -1 | sed 's/:.*//')
echo "kernel: $NAME"
awk -v n="$NAME:" '$1==n{p=1} p{print} p&&/^\.Lfunc_end/{exit}' /tmp/all.s > /tmp/isa.s
wc -l /tmp/isa.s
grep -n "^.LBB\|s_cbranch\|s_branch\|s_barrier\|scratch_" /tmp/isa.s
