"""ctypes loader for the CPU parity oracle (oracle/liborc.so, oracle/_ref/libhexlfpga_ref.so).

TEST INFRASTRUCTURE ONLY. Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package (hexl-fpga_amd/).
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liborc.so"
REF_LIB = HERE / "_ref" / "libhexlfpga_ref.so"
REFERENCE_TREE = Path("/root/reference")

u64 = ctypes.c_uint64
P = ctypes.POINTER(u64)
sz = ctypes.c_size_t


def build(with_ref: bool = True):
    subprocess.run(["make", "-C", str(HERE), "-s"], check=True)
    if with_ref and REFERENCE_TREE.exists():
        subprocess.run(["make", "-C", str(HERE), "-s", "ref"], check=True)


def p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P)


_orc = None
_ref = None


def orc() -> ctypes.CDLL:
    global _orc
    if _orc is None:
        if not LIB.exists():
            build(with_ref=False)
        L = ctypes.CDLL(str(LIB))
        L.orc_mulmod.restype = u64; L.orc_mulmod.argtypes = [u64, u64, u64]
        L.orc_powmod.restype = u64; L.orc_powmod.argtypes = [u64, u64, u64]
        L.orc_invmod.restype = u64; L.orc_invmod.argtypes = [u64, u64]
        L.orc_is_prime.restype = ctypes.c_int; L.orc_is_prime.argtypes = [u64]
        L.orc_generate_primes.restype = sz; L.orc_generate_primes.argtypes = [P, sz, ctypes.c_uint, u64]
        L.orc_minimal_primitive_root.restype = u64; L.orc_minimal_primitive_root.argtypes = [u64, u64]
        L.orc_shoup_factor.restype = u64; L.orc_shoup_factor.argtypes = [u64, u64]
        L.orc_tables_hexl.restype = None; L.orc_tables_hexl.argtypes = [u64, u64, u64, P, P, P, P]
        L.orc_tables_keyswitch.restype = None; L.orc_tables_keyswitch.argtypes = [u64, u64, u64, P]
        L.orc_ntt_fwd.restype = None; L.orc_ntt_fwd.argtypes = [P, u64, u64, P, P]
        L.orc_ntt_inv.restype = None; L.orc_ntt_inv.argtypes = [P, u64, u64, P, P, u64, u64]
        L.orc_dyadic_multiply.restype = None
        L.orc_dyadic_multiply.argtypes = [P, P, P, u64, P, u64, ctypes.c_int]
        L.orc_keyswitch.restype = ctypes.c_int
        L.orc_keyswitch.argtypes = [P, P, u64, u64, u64, u64, u64, P, ctypes.POINTER(P), P, P]
        L.orc_ks_ntt.restype = None; L.orc_ks_ntt.argtypes = [P, u64, u64, P]
        L.orc_ks_intt.restype = None; L.orc_ks_intt.argtypes = [P, u64, u64, P]
        L.orc_fnv1a64.restype = u64; L.orc_fnv1a64.argtypes = [ctypes.c_void_p, sz]
        L.orc_fill_splitmix.restype = None; L.orc_fill_splitmix.argtypes = [P, sz, u64, u64]
        _orc = L
    return _orc


def ref():
    """The reference's own CPU code (built from /root/reference by `make -C oracle ref`); None if absent."""
    global _ref
    if _ref is None:
        if not REF_LIB.exists():
            return None
        L = ctypes.CDLL(str(REF_LIB))
        L.ref_ntt_tables.restype = u64; L.ref_ntt_tables.argtypes = [u64, u64, P, P, P, P]
        L.ref_ntt_forward.restype = None; L.ref_ntt_forward.argtypes = [P, P, u64, u64]
        L.ref_ntt_inverse.restype = None; L.ref_ntt_inverse.argtypes = [P, P, u64, u64]
        L.ref_fwd_with_tables.restype = None; L.ref_fwd_with_tables.argtypes = [P, u64, u64, P, P]
        L.ref_inv_with_tables.restype = None; L.ref_inv_with_tables.argtypes = [P, u64, u64, P, P]
        L.ref_generate_primes.restype = sz; L.ref_generate_primes.argtypes = [P, sz, sz, sz]
        L.ref_minimal_primitive_root.restype = u64; L.ref_minimal_primitive_root.argtypes = [u64, u64]
        L.ref_inverse_mod.restype = u64; L.ref_inverse_mod.argtypes = [u64, u64]
        L.ref_ks_tables.restype = None; L.ref_ks_tables.argtypes = [u64, u64, P]
        _ref = L
    return _ref


# ---------------------------------------------------------------- numpy-level conveniences
def primes(num: int, bits: int, n: int) -> list[int]:
    out = np.zeros(num, dtype=np.uint64)
    got = orc().orc_generate_primes(p(out), num, bits, n)
    assert got == num
    return [int(v) for v in out]


def splitmix(count: int, seed: int, q: int = 0) -> np.ndarray:
    x = np.empty(count, dtype=np.uint64)
    orc().orc_fill_splitmix(p(x), count, seed, q)
    return x


def fnv(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return orc().orc_fnv1a64(a.ctypes.data, a.nbytes)


class HexlTables:
    """HEXL-layout tables for the standalone NTT/INTT (tests/test_utils/ntt.cpp:290-384)."""

    def __init__(self, n: int, q: int, w: int | None = None):
        self.n, self.q = n, q
        self.w = orc().orc_minimal_primitive_root(2 * n, q) if w is None else w
        self.roots, self.precon, self.inv_roots, self.inv_precon = (np.zeros(n, dtype=np.uint64) for _ in range(4))
        orc().orc_tables_hexl(n, q, self.w, p(self.roots), p(self.precon), p(self.inv_roots), p(self.inv_precon))
        self.inv_n = orc().orc_invmod(n, q)
        self.inv_n_w = orc().orc_mulmod(self.inv_n, int(self.inv_roots[n - 1]), q)   # test_inv_ntt.cpp:107-111


def ntt_fwd(x: np.ndarray, t: HexlTables) -> np.ndarray:
    y = np.ascontiguousarray(x, dtype=np.uint64).copy().reshape(-1, t.n)   # always (batch, n)
    for row in y:
        orc().orc_ntt_fwd(p(row), t.n, t.q, p(t.roots), p(t.precon))
    return y


def ntt_inv(x: np.ndarray, t: HexlTables) -> np.ndarray:
    y = np.ascontiguousarray(x, dtype=np.uint64).copy().reshape(-1, t.n)   # always (batch, n)
    for row in y:
        orc().orc_ntt_inv(p(row), t.n, t.q, p(t.inv_roots), p(t.inv_precon), t.inv_n, t.inv_n_w)
    return y


def dyadic(a: np.ndarray, b: np.ndarray, n: int, moduli: np.ndarray, exact: bool = True) -> np.ndarray:
    nm = len(moduli)
    out = np.empty(3 * nm * n, dtype=np.uint64)
    orc().orc_dyadic_multiply(p(out), p(a), p(b), n, p(np.ascontiguousarray(moduli, dtype=np.uint64)), nm,
                              1 if exact else 0)
    return out


def keyswitch(result: np.ndarray, t_target: np.ndarray, n, L, K, rns, moduli, keys, modswitch, twiddles=None):
    """in place on `result` (accumulates), like intel::hexl::KeySwitch"""
    karr = (P * len(keys))(*[p(k) for k in keys])
    rc = orc().orc_keyswitch(p(result), p(t_target), n, L, K, rns, 2,
                             p(np.ascontiguousarray(moduli, dtype=np.uint64)), karr,
                             p(np.ascontiguousarray(modswitch, dtype=np.uint64)),
                             None if twiddles is None else p(twiddles))
    assert rc == 0
    return result


# ---------------------------------------------------------------- bench.py's CPU leg (oracle/cpu_baseline.c)
_cb = {}


def cpu_baseline_lib(march: str = "native") -> ctypes.CDLL:
    """build (on THIS host, -march=native by default) and load the OpenMP CPU port that bench.py times"""
    if march not in _cb:
        if march == "native":                                  # never trust a shipped "native" build: rebuild here
            (HERE / "_native" / "libcpubase.native.so").unlink(missing_ok=True)
        subprocess.run(["make", "-C", str(HERE), "-s", "cpubase", f"MARCH={march}"], check=True)
        L = ctypes.CDLL(str(HERE / "_native" / f"libcpubase.{march}.so"))
        L.cb_plan_create.restype = ctypes.c_void_p
        L.cb_plan_create.argtypes = [u64, u64, u64, P, ctypes.POINTER(P), P]
        L.cb_plan_destroy.restype = None; L.cb_plan_destroy.argtypes = [ctypes.c_void_p]
        L.cb_keyswitch_batch.restype = ctypes.c_int
        L.cb_keyswitch_batch.argtypes = [ctypes.c_void_p, P, P, u64, ctypes.c_int]
        L.cb_keyswitch_timed.restype = u64
        L.cb_keyswitch_timed.argtypes = [ctypes.c_void_p, P, P, u64, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_int), P]
        L.cb_stream_triad.restype = ctypes.c_double
        L.cb_stream_triad.argtypes = [ctypes.c_int, ctypes.c_double, u64]
        L.cb_ntt_fwd_batch.restype = ctypes.c_int
        L.cb_ntt_fwd_batch.argtypes = [ctypes.c_void_p, u64, P, u64, ctypes.c_int]
        L.cb_max_threads.restype = ctypes.c_int; L.cb_max_threads.argtypes = []
        L.cb_isa.restype = ctypes.c_char_p; L.cb_isa.argtypes = [ctypes.c_void_p]
        _cb[march] = L
    return _cb[march]


class CpuKeySwitch:
    """tables + key Shoup factors precomputed once; keyswitch_batch(results[b][2][L][n], ts[b][L][n], threads)"""

    def __init__(self, n, L, K, moduli, keys, modswitch, march="native"):
        self.lib = cpu_baseline_lib(march)
        self.n, self.L, self.K = n, L, K
        self._keys = [np.ascontiguousarray(k, dtype=np.uint64) for k in keys]          # keep alive
        self._karr = (P * len(keys))(*[p(k) for k in self._keys])
        self._mod = np.ascontiguousarray(moduli, dtype=np.uint64)
        self._msf = np.ascontiguousarray(modswitch, dtype=np.uint64)
        self.h = self.lib.cb_plan_create(n, L, K, p(self._mod), self._karr, p(self._msf))

    def keyswitch_batch(self, results: np.ndarray, ts: np.ndarray, threads: int = 0) -> int:
        batch = ts.size // (self.L * self.n)
        assert results.size == batch * 2 * self.L * self.n
        return self.lib.cb_keyswitch_batch(self.h, p(results), p(ts), batch, threads)

    def keyswitch_timed(self, ts: np.ndarray, rs: np.ndarray, threads: int, seconds: float):
        """the benchmark leg: pinned threads, private first-touched ciphertexts, keys / tables replicated per NUMA node; returns
        (keyswitches done, seconds of the slowest thread, NUMA nodes used, thread 0's first result after one keyswitch)"""
        nsrc = ts.size // (self.L * self.n)
        assert rs.size == nsrc * 2 * self.L * self.n
        el, nodes = ctypes.c_double(0), ctypes.c_int(0)
        first = np.empty(2 * self.L * self.n, dtype=np.uint64)
        done = self.lib.cb_keyswitch_timed(self.h, p(ts), p(rs), nsrc, threads, seconds, ctypes.byref(el), ctypes.byref(nodes), p(first))
        return int(done), el.value, nodes.value, first

    def isa(self) -> str:
        """kernels the plan's first modulus runs on: scalar / avx512dq / avx512ifma (HEXL's rule; HEXL_CPU_ISA restricts)"""
        return self.lib.cb_isa(self.h).decode()

    def ntt_fwd_batch(self, x: np.ndarray, modulus_index: int = 0, threads: int = 0) -> int:
        return self.lib.cb_ntt_fwd_batch(self.h, modulus_index, p(x), x.size // self.n, threads)

    def close(self):
        if self.h:
            self.lib.cb_plan_destroy(self.h)
            self.h = None
