#!/bin/bash
# sha256 of the DEVICE code (the .hip_fatbin section: the gfx950 code object bundle) of every kernel object of the library.
# Host code carries __LINE__ (HX_CHECK), so an object's own hash moves whenever a line above a launcher moves; the kernels do not.
# usage: tools/device_code_sha.sh [objdir]   (default hexl-fpga_amd/lib)
dir=${1:-$(dirname "$0")/../hexl-fpga_amd/lib}
for o in ntt dyadic keyswitch keyswitch_f64 keyswitch_lat keyswitch_x capi; do
    t=$(mktemp)
    /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$t" "$dir/$o.o" /dev/null 2>/dev/null || { echo "no .hip_fatbin in $o.o"; continue; }
    echo "$(sha256sum < "$t" | cut -c1-64)  $o.o:.hip_fatbin ($(stat -c %s "$t") bytes)"
    rm -f "$t"
done
