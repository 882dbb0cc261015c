#!/bin/bash
# build_variant.sh <name> <flags...> -- libhexl_mi355x.so with keyswitch_x.hip compiled under extra -D flags, into
# hexl-fpga_amd/lib_var/<name>/ (git-ignored; travels with gpurun). Select it with HEXL_MI355X_LIB=<path>.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/hexl-fpga_amd/csrc; O=$R/hexl-fpga_amd/lib_var/$NAME
mkdir -p $O
make -C $C -j4 -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c $C/keyswitch_x.hip -o $O/keyswitch_x.o
L=$R/hexl-fpga_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libhexl_mi355x.so -o $O/libhexl_mi355x.so $L/ntt.o $L/dyadic.o $L/keyswitch.o $L/keyswitch_f64.o $L/keyswitch_lat.o $O/keyswitch_x.o $L/capi.o
rm -f $O/keyswitch_x.o
echo "built $O/libhexl_mi355x.so ($*)"
