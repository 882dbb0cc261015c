"""hexl-fpga_amd -- MI355X-native FHE primitives behind the hexl-fpga interface.

Python side of the package: a thin ctypes binding over the C-ABI of
``include/hexl_mi355x.h`` (``lib/libhexl_mi355x.so``, hand-written HIP for gfx950) plus
``HexlFpga``, a host-side mirror of the reference's public API
(``host/inc/hexl-fpga.h:15-161``: ``set_worksize_X / X / XCompleted``) used by the parity
tests so they read like the reference's own gtests.

torch is plumbing only: device buffers, streams and ``torch.distributed``. All arithmetic
happens in the HIP kernels; there is NO CPU fallback -- if the extension is missing or no
gfx950 device is visible every entry point raises.

The directory name contains a hyphen (it mirrors the reference's project name), so import
it through the repo-root shim: ``import hexl_fpga_amd``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
ROOT = _PKG.parent
# HEXL_MI355X_LIB: another build of the same library (tools/build_variant.sh: kernel experiments side by side on one box)
LIB_PATH = Path(os.environ["HEXL_MI355X_LIB"]) if os.environ.get("HEXL_MI355X_LIB") else _PKG / "lib" / "libhexl_mi355x.so"

_u64 = ctypes.c_uint64
_sz = ctypes.c_size_t
_vp = ctypes.c_void_p
_i = ctypes.c_int

# symbol -> argtypes; every function returns int status. Mirrors include/hexl_mi355x.h 1:1
# (tests/test_abi.py checks the header and this table against the built library).
C_ABI = {
    "hexl_device_count": [],
    "hexl_ctx_create": [_i, ctypes.POINTER(_vp)],
    "hexl_ctx_destroy": [_vp],
    "hexl_ctx_set_stream": [_vp, _vp],
    "hexl_ctx_use_own_stream": [_vp],
    "hexl_ctx_sync": [_vp],
    "hexl_ctx_describe": [_vp, ctypes.c_char_p, _sz],
    "hexl_ntt_fwd": [_vp, _vp, _sz, _vp, _vp, _u64, _u64],
    "hexl_ntt_inv": [_vp, _vp, _sz, _vp, _vp, _u64, _u64, _u64, _u64],
    "hexl_dyadic_multiply": [_vp, _vp, _vp, _vp, _sz, _u64, _vp, _u64],
    "hexl_ks_plan_create": [_vp, _u64, _u64, _u64, _u64, _u64, _vp, _vp, _vp, ctypes.POINTER(_vp)],
    "hexl_ks_plan_destroy": [_vp],
    "hexl_ks_set_keys": [_vp, ctypes.POINTER(_vp)],
    "hexl_keyswitch": [_vp, _vp, _vp, _sz],
    "hexl_ks_range_check": [_vp],
    "hexl_ks_plan_tiers": [_vp, ctypes.POINTER(_i)],
    "hexl_multiply_relinearize": [_vp, _vp, _vp, _vp, _sz],
    "hexl_ks_scratch_bytes": [_vp, _sz],
    "hexl_ntt_fwd_host": [_vp, ctypes.POINTER(_vp), _sz, _vp, _vp, _u64, _u64],
    "hexl_ntt_inv_host": [_vp, ctypes.POINTER(_vp), _sz, _vp, _vp, _u64, _u64, _u64, _u64],
    "hexl_dyadic_multiply_host": [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _sz, _u64,
                                  ctypes.POINTER(_vp), _u64],
    "hexl_keyswitch_host": [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _sz],
    "hexl_ks_time_stages": [_vp, _vp, _vp, _sz, _i, ctypes.POINTER(ctypes.c_float)],
}


class HexlError(RuntimeError):
    pass


def build(force: bool = False) -> Path:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_PKG / "csrc"), "-j4"] + (["-B"] if force else []), check=True)
    else:
        # cheap staleness check: make decides
        subprocess.run(["make", "-C", str(_PKG / "csrc"), "-j4", "-s"], check=True)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    """Load libhexl_mi355x.so; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise HexlError(f"{LIB_PATH} missing: run __graft_entry__.build() / make -C hexl-fpga_amd/csrc "
                            "(there is no CPU fallback)")
        # torch must load its HIP runtime first so this library binds to the SAME libamdhip64 instance
        # (device pointers from torch tensors are only meaningful inside one runtime).
        import torch  # noqa: F401
        _lib = ctypes.CDLL(str(LIB_PATH))
        for name, args in C_ABI.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = _sz if name == "hexl_ks_scratch_bytes" else _i
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise HexlError(f"{what} failed with status {rc}")


def ptr_array(arrays):
    """ctypes array of pointers to numpy arrays (host) or torch tensors (device), kept alive by the caller"""
    return (_vp * len(arrays))(*[_ptr(a) for a in arrays])


def _ptr(t) -> int:
    """device pointer of a torch tensor / host pointer of a numpy array"""
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


def as_i64(a: np.ndarray):
    """numpy uint64 -> torch int64 view (bit pattern preserved)"""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64))


def to_u64(t) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint64)


class Context:
    """One GPU: stream + scratch (hexl_ctx). Device-pointer launchers take torch int64 CUDA tensors."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        import torch
        if not torch.cuda.is_available():
            raise HexlError("no ROCm device visible: the HIP path is the only path")
        self.device = device
        h = _vp()
        _check(lib().hexl_ctx_create(device, ctypes.byref(h)), "hexl_ctx_create")
        self.h = h
        if use_torch_stream:
            with torch.cuda.device(device):
                self.set_stream(torch.cuda.current_stream().cuda_stream)

    def set_stream(self, raw_stream: int):
        _check(lib().hexl_ctx_set_stream(self.h, _vp(raw_stream)), "hexl_ctx_set_stream")

    def sync(self):
        _check(lib().hexl_ctx_sync(self.h), "hexl_ctx_sync")

    def describe(self) -> str:
        buf = ctypes.create_string_buffer(256)
        _check(lib().hexl_ctx_describe(self.h, buf, 256), "hexl_ctx_describe")
        return buf.value.decode()

    def close(self):
        if getattr(self, "h", None):
            lib().hexl_ctx_destroy(self.h)
            self.h = None

    # ---- K1 / K2 / K3 on device tensors (in place / out of place as the reference) ----
    def ntt_fwd(self, x, roots, precon, q: int, n: int):
        batch = x.numel() // n
        _check(lib().hexl_ntt_fwd(self.h, _ptr(x), batch, _ptr(roots), _ptr(precon), q, n), "hexl_ntt_fwd")

    def ntt_inv(self, x, inv_roots, inv_precon, q: int, inv_n: int, inv_n_w: int, n: int):
        batch = x.numel() // n
        _check(lib().hexl_ntt_inv(self.h, _ptr(x), batch, _ptr(inv_roots), _ptr(inv_precon), q, inv_n, inv_n_w, n),
               "hexl_ntt_inv")

    def dyadic_multiply(self, out, a, b, moduli, n: int, n_moduli: int):
        batch = a.numel() // (2 * n_moduli * n)
        _check(lib().hexl_dyadic_multiply(self.h, _ptr(out), _ptr(a), _ptr(b), batch, n, _ptr(moduli), n_moduli),
               "hexl_dyadic_multiply")


class KeySwitchPlan:
    """Device state for one keyswitch parameter set (hexl_ks_plan): tables, constants, keys."""

    def __init__(self, ctx: Context, n: int, L: int, K: int, rns: int, kcc: int, moduli, modswitch,
                 twiddles=None):
        self.ctx, self.n, self.L, self.K = ctx, n, L, K
        mod = np.ascontiguousarray(moduli, dtype=np.uint64)
        msf = np.ascontiguousarray(modswitch, dtype=np.uint64)
        tw = None if twiddles is None else np.ascontiguousarray(twiddles, dtype=np.uint64)
        h = _vp()
        _check(lib().hexl_ks_plan_create(ctx.h, n, L, K, rns, kcc, mod.ctypes.data, msf.ctypes.data,
                                         None if tw is None else tw.ctypes.data, ctypes.byref(h)),
               "hexl_ks_plan_create")
        self.h = h

    def set_keys(self, keys):
        """keys: list of L numpy uint64 arrays, keys[d][(k*K + i)*n + j]"""
        self._keys = [np.ascontiguousarray(k, dtype=np.uint64) for k in keys]
        arr = (_vp * len(self._keys))(*[k.ctypes.data for k in self._keys])
        _check(lib().hexl_ks_set_keys(self.h, arr), "hexl_ks_set_keys")

    def keyswitch(self, result, t_target, batch: int):
        _check(lib().hexl_keyswitch(self.h, _ptr(result), _ptr(t_target), batch), "hexl_keyswitch")

    def range_check(self) -> bool:
        """True if every keyswitch since the last check saw in-range words (syncs the stream, clears the flag)"""
        rc = lib().hexl_ks_range_check(self.h)
        if rc not in (0, 1):                                       # 1 = HEXL_W_RANGE
            _check(rc, "hexl_ks_range_check")
        return rc == 0

    def tiers(self):
        """(per-limb forward reduction periods [K] -- 0 = strict, -1 = integer kernels --, True when the limbs in use differ)"""
        out = (_i * self.K)()
        rc = lib().hexl_ks_plan_tiers(self.h, out)
        if rc not in (0, 1):
            _check(rc, "hexl_ks_plan_tiers")
        return list(out), rc == 1

    def multiply_relinearize(self, out, a, b, batch: int):
        """out[batch][2][L][n] = (a0 b0, a0 b1 + a1 b0) + KeySwitch(a1 b1), one fused pass (N = 1024 ... 16384)"""
        _check(lib().hexl_multiply_relinearize(self.h, _ptr(out), _ptr(a), _ptr(b), batch), "hexl_multiply_relinearize")

    def keyswitch_host(self, results, t_targets):
        n = len(results)
        r = (_vp * n)(*[_ptr(a) for a in results])
        t = (_vp * n)(*[_ptr(a) for a in t_targets])
        rc = lib().hexl_keyswitch_host(self.h, r, t, n)
        if rc != 1:                                                # 1 = HEXL_W_RANGE: computed, some object had an out-of-range word
            _check(rc, "hexl_keyswitch_host")
        return rc == 0

    def time_stages(self, result, t_target, batch: int, iters: int):
        out = (ctypes.c_float * 4)()
        _check(lib().hexl_ks_time_stages(self.h, _ptr(result), _ptr(t_target), batch, iters, out),
               "hexl_ks_time_stages")
        return list(out)

    def scratch_bytes(self, batch: int) -> int:
        return lib().hexl_ks_scratch_bytes(self.h, batch)

    def close(self):
        if getattr(self, "h", None):
            lib().hexl_ks_plan_destroy(self.h)
            self.h = None


from .host_api import HexlFpga  # noqa: E402  (mirror of host/inc/hexl-fpga.h)

__all__ = ["Context", "KeySwitchPlan", "HexlFpga", "HexlError", "build", "lib", "as_i64", "to_u64", "C_ABI",
           "LIB_PATH", "ROOT"]
