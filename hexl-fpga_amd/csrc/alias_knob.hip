// alias_knob.hip -- the one function that decides whether the slot-major keyswitch launcher may alias a stream (keyswitch_x.hip
// ksx_alias_mask). The shipped library (libhexl_mi355x.so) is built WITHOUT -DHEXL_KEY_ALIAS_KNOB: the answer is a constant 0, no
// environment variable is read. libhexl_mi355x_keyalias.so links the SAME kernel objects with this file built -DHEXL_KEY_ALIAS_KNOB:
// HEXL_KSX_ALIAS=1 then makes every key row read row 0 (key_stride = 0; WRONG results by design), which takes the key stream out of the
// L2-miss-side counters while every kernel instruction stays what the shipped library runs -- what bench.py's key-stream figure needs.
#include <stdio.h>
#include <stdlib.h>

#include "hexl_internal.hpp"

u32 hx_ksx_alias_mask() {
#ifdef HEXL_KEY_ALIAS_KNOB
    static const u32 mask = [] {
        const char* e = getenv("HEXL_KSX_ALIAS");
        const u32 m = (e ? (u32)atoi(e) : 0u) & 1u;                // key rows only: the other streams need kernel support (profiling build)
        if (m) fprintf(stderr, "[hexl_mi355x KEY-ALIAS BUILD] HEXL_KSX_ALIAS=1: every key row reads row 0 -- keyswitch results are WRONG by design\n");
        return m;
    }();
    return mask;
#else
    return 0u;
#endif
}
