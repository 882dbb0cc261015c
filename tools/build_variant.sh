#!/bin/bash
# build_variant.sh <name> <flags...> -- libhexl_mi355x.so with some translation units (VAR_FILES, default "keyswitch_x") compiled
# under extra -D flags, into hexl-fpga_amd/lib_var/<name>/ (git-ignored; travels with gpurun). Select it with
# HEXL_MI355X_LIB=<path>. Run `make -C hexl-fpga_amd/csrc` first: the other objects come from hexl-fpga_amd/lib.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/hexl-fpga_amd/csrc; O=$R/hexl-fpga_amd/lib_var/$NAME; L=$R/hexl-fpga_amd/lib
mkdir -p $O
OBJS=""
for f in ntt dyadic keyswitch keyswitch_f64 keyswitch_lat keyswitch_x capi; do
  if [[ " ${VAR_FILES:-keyswitch_x} " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c $C/$f.hip -o $O/$f.o
    OBJS="$OBJS $O/$f.o"
  else
    OBJS="$OBJS $L/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libhexl_mi355x.so -o $O/libhexl_mi355x.so $OBJS $L/alias_knob.o $L/host_simd.o
rm -f $O/*.o
echo "built $O/libhexl_mi355x.so ($*)"
