"""Host-side mirror of the reference's public API (host/inc/hexl-fpga.h:15-161).

Same names, argument order and batching contract as ``intel::hexl::*``:
``set_worksize_X(ws)``, then ``ws`` calls of ``X(...)``, then ``XCompleted()`` (which always
returns True and resets the worksize to 1, host/src/fpga_int.cpp:209-232,484-507). With
worksize 1 a call completes before it returns (fpga_int.cpp:459-461). A batch never spans a
change of modulus (NTT/INTT, fpga_int.cpp:346-353) or of keyswitch parameters / key pointers
(fpga_int.cpp:429-447): such a change acts as a fence and flushes what is queued.

Arrays are numpy uint64, caller-owned and updated in place exactly where the reference
writes through the caller's pointers (``_NTT``/``_INTT`` operand, ``KeySwitch`` result,
``DyadicMultiply`` results). Everything runs on the GPU through the C-ABI host-pointer entry
points; argument errors raise ``ValueError`` where the reference FPGA_ASSERTs
(host/src/{ntt,intt,keyswitch,dyadic_multiply}.cpp).
"""
from __future__ import annotations

import numpy as np


def _u64(a, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError(f"{name} must be a C-contiguous numpy uint64 array")
    return a


class HexlFpga:
    def __init__(self):
        self._ctx = None
        self._ws = {"dyadic": 1, "ks": 1, "ntt": 1, "intt": 1}
        self._q = {"dyadic": [], "ks": [], "ntt": [], "intt": []}
        self._plans = {}

    # ---- resources (host/src/fpga_context.cpp:15-25) ----
    def acquire_FPGA_resources(self, device: int = 0):
        from . import Context
        if self._ctx is None:
            self._ctx = Context(device, use_torch_stream=False)

    def release_FPGA_resources(self):
        for p in self._plans.values():
            p.close()
        self._plans.clear()
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def _need(self):
        if self._ctx is None:
            raise RuntimeError("acquire_FPGA_resources() has not been called")
        return self._ctx

    # ---- DyadicMultiply (host/src/dyadic_multiply.cpp:15-26) ----
    def set_worksize_DyadicMultiply(self, ws: int):
        self._ws["dyadic"] = int(ws)

    def DyadicMultiply(self, results, operand1, operand2, n: int, moduli, n_moduli: int):
        if n <= 0 or n_moduli <= 0:                       # host/src/dyadic_multiply.cpp:19-21
            raise ValueError("n and n_moduli must be positive integers")
        _u64(results, "results"), _u64(operand1, "operand1"), _u64(operand2, "operand2"), _u64(moduli, "moduli")
        q = self._q["dyadic"]
        if q and (q[0][3], q[0][5]) != (n, n_moduli):
            self._flush_dyadic()
        q.append((results, operand1, operand2, n, moduli, n_moduli))
        if self._ws["dyadic"] == 1:
            self.DyadicMultiplyCompleted()

    def DyadicMultiplyCompleted(self) -> bool:
        self._flush_dyadic()
        self._ws["dyadic"] = 1
        return True

    def _flush_dyadic(self):
        from . import lib, _check
        q = list(self._q["dyadic"])
        self._q["dyadic"].clear()      # in place: callers hold a reference to the live queue
        if not q:
            return
        ctx = self._need()
        from . import ptr_array
        n, nm = q[0][3], q[0][5]
        _check(lib().hexl_dyadic_multiply_host(ctx.h, ptr_array([o[0] for o in q]), ptr_array([o[1] for o in q]),
                                               ptr_array([o[2] for o in q]), len(q), n,
                                               ptr_array([o[4] for o in q]), nm), "hexl_dyadic_multiply_host")

    # ---- _NTT / _INTT (host/src/ntt.cpp:15-28, intt.cpp:15-29) ----
    def _set_worksize_NTT(self, ws: int):
        self._ws["ntt"] = int(ws)

    def _NTT(self, operand, root_of_unity_powers, precon_root_of_unity_powers, coeff_modulus: int, n: int):
        if n not in (1024, 2048, 4096, 8192, 16384):      # reference: 16384 only
            raise ValueError("requires n = 16384 (extended here to 1024..16384)")
        _u64(operand, "operand"), _u64(root_of_unity_powers, "roots"), _u64(precon_root_of_unity_powers, "precon")
        q = self._q["ntt"]
        if q and (q[0][3], q[0][4]) != (coeff_modulus, n):      # fence on modulus change
            self._flush_ntt()
        q.append((operand, root_of_unity_powers, precon_root_of_unity_powers, coeff_modulus, n))
        if self._ws["ntt"] == 1:
            self._NTTCompleted()

    def _NTTCompleted(self) -> bool:
        self._flush_ntt()
        self._ws["ntt"] = 1
        return True

    def _flush_ntt(self):
        from . import lib, _check
        q = list(self._q["ntt"])
        self._q["ntt"].clear()      # in place: callers hold a reference to the live queue
        if not q:
            return
        ctx = self._need()
        _, roots, precon, mod, n = q[0]                       # tables of the first object (fpga.cpp:403-411)
        from . import ptr_array
        _check(lib().hexl_ntt_fwd_host(ctx.h, ptr_array([o[0] for o in q]), len(q), roots.ctypes.data,
                                       precon.ctypes.data, mod, n), "hexl_ntt_fwd_host")

    def _set_worksize_INTT(self, ws: int):
        self._ws["intt"] = int(ws)

    def _INTT(self, operand, inv_root_of_unity_powers, precon_inv_root_of_unity_powers, coeff_modulus: int,
              inv_n: int, inv_n_w: int, n: int):
        if n not in (1024, 2048, 4096, 8192, 16384):
            raise ValueError("requires n = 16384 (extended here to 1024..16384)")
        _u64(operand, "operand"), _u64(inv_root_of_unity_powers, "inv_roots")
        _u64(precon_inv_root_of_unity_powers, "inv_precon")
        q = self._q["intt"]
        if q and (q[0][3], q[0][6]) != (coeff_modulus, n):
            self._flush_intt()
        q.append((operand, inv_root_of_unity_powers, precon_inv_root_of_unity_powers, coeff_modulus, inv_n,
                  inv_n_w, n))
        if self._ws["intt"] == 1:
            self._INTTCompleted()

    def _INTTCompleted(self) -> bool:
        self._flush_intt()
        self._ws["intt"] = 1
        return True

    def _flush_intt(self):
        from . import lib, _check
        q = list(self._q["intt"])
        self._q["intt"].clear()      # in place: callers hold a reference to the live queue
        if not q:
            return
        ctx = self._need()
        _, ir, ip, mod, inv_n, inv_n_w, n = q[0]
        from . import ptr_array
        _check(lib().hexl_ntt_inv_host(ctx.h, ptr_array([o[0] for o in q]), len(q), ir.ctypes.data, ip.ctypes.data,
                                       mod, inv_n, inv_n_w, n), "hexl_ntt_inv_host")

    # ---- KeySwitch (host/src/keyswitch.cpp:15-41) ----
    def set_worksize_KeySwitch(self, ws: int):
        self._ws["ks"] = int(ws)

    def KeySwitch(self, result, t_target_iter_ptr, n: int, decomp_modulus_size: int, key_modulus_size: int,
                  rns_modulus_size: int, key_component_count: int, moduli, k_switch_keys, modswitch_factors,
                  twiddle_factors=None):
        if n not in (1024, 2048, 4096, 8192, 16384):
            raise ValueError("requires n = 16384/8192/4096/2048/1024")
        if decomp_modulus_size <= 0 or rns_modulus_size <= 0:
            raise ValueError("requires decomp_modulus_size > 0 and rns_modulus_size > 0")
        if key_component_count != 2:
            raise ValueError("requires key_component_count = 2")
        if not (decomp_modulus_size < key_modulus_size <= 16):       # reference: key_modulus_size <= 7
            raise ValueError("requires decomp_modulus_size < key_modulus_size <= 16")
        _u64(result, "result"), _u64(t_target_iter_ptr, "t_target_iter_ptr"), _u64(moduli, "moduli")
        _u64(modswitch_factors, "modswitch_factors")
        sig = (n, decomp_modulus_size, key_modulus_size, rns_modulus_size, tuple(moduli[:key_modulus_size].tolist()),
               tuple(modswitch_factors[:key_modulus_size].tolist()),
               tuple(k.ctypes.data for k in k_switch_keys[:decomp_modulus_size]))   # key pointer identity
        q = self._q["ks"]
        if q and q[0][0] != sig:                                  # fence (fpga_int.cpp:429-447)
            self._flush_ks()
        q.append((sig, result, t_target_iter_ptr, moduli, k_switch_keys, modswitch_factors, twiddle_factors))
        if self._ws["ks"] == 1:
            self.KeySwitchCompleted()

    def KeySwitchCompleted(self) -> bool:
        self._flush_ks()
        self._ws["ks"] = 1
        return True

    def _flush_ks(self):
        from . import KeySwitchPlan
        q = list(self._q["ks"])
        self._q["ks"].clear()      # in place: callers hold a reference to the live queue
        if not q:
            return
        ctx = self._need()
        sig, _, _, moduli, keys, msf, tw = q[0]
        n, L, K, rns = sig[0], sig[1], sig[2], sig[3]
        plan = self._plans.get(sig)
        if plan is None:                                          # device key / twiddle cache (fpga.cpp:1158-1165)
            plan = KeySwitchPlan(ctx, n, L, K, rns, 2, moduli[:K], msf[:K], tw)
            plan.set_keys([k[: 2 * K * n] for k in keys[:L]])
            self._plans[sig] = plan
        plan.keyswitch_host([o[1] for o in q], [o[2] for o in q])
