"""CPU: the C-ABI library loads and exports every symbol include/hexl_mi355x.h declares (no compute calls),
the ctypes table covers the header 1:1, and libhexl-fpga.so exports the reference's 14 mangled C++ symbols
(SURVEY 8b)."""
import ctypes
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "hexl_mi355x.h").read_text()
DECLARED = sorted(set(re.findall(r"^(?:int|size_t)\s+(hexl_\w+)\s*\(", HEADER, flags=re.M)))

REFERENCE_CXX_SYMBOLS = [
    "_ZN5intel4hexl22acquire_FPGA_resourcesEv", "_ZN5intel4hexl22release_FPGA_resourcesEv",
    "_ZN5intel4hexl27set_worksize_DyadicMultiplyEm", "_ZN5intel4hexl14DyadicMultiplyEPmPKmS3_mS3_m",
    "_ZN5intel4hexl23DyadicMultiplyCompletedEv", "_ZN5intel4hexl22set_worksize_KeySwitchEm",
    "_ZN5intel4hexl9KeySwitchEPmPKmmmmmmS3_PS3_S3_S3_", "_ZN5intel4hexl18KeySwitchCompletedEv",
    "_ZN5intel4hexl17_set_worksize_NTTEm", "_ZN5intel4hexl4_NTTEPmPKmS3_mm", "_ZN5intel4hexl13_NTTCompletedEv",
    "_ZN5intel4hexl18_set_worksize_INTTEm", "_ZN5intel4hexl5_INTTEPmPKmS3_mmmm", "_ZN5intel4hexl14_INTTCompletedEv",
]


def test_header_declares_the_expected_surface():
    assert len(DECLARED) >= 18
    for must in ("hexl_ctx_create", "hexl_ntt_fwd", "hexl_ntt_inv", "hexl_dyadic_multiply", "hexl_ks_plan_create",
                 "hexl_ks_set_keys", "hexl_keyswitch", "hexl_keyswitch_host"):
        assert must in DECLARED


def test_library_exports_every_declared_symbol(hx):
    hx.build()
    lib = ctypes.CDLL(str(hx.LIB_PATH))
    for name in DECLARED:
        assert hasattr(lib, name), f"{name} declared in include/hexl_mi355x.h but not exported"
    assert sorted(hx.C_ABI) == DECLARED, "ctypes table and header disagree"
    hx.lib()        # resolves argtypes for all of them


def test_no_cpu_fallback_without_gpu(hx):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hx.HexlError):
        hx.Context(0)


def test_cxx_api_symbols():
    so = ROOT / "hexl-fpga_amd" / "lib" / "libhexl-fpga.so"
    if not so.exists():
        subprocess.run(["make", "-C", str(ROOT / "hexl-fpga_amd" / "host")], check=True)
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for sym in REFERENCE_CXX_SYMBOLS:
        assert sym in exported, sym


def test_product_never_touches_the_oracle():
    """the shipped package must not import / link / call anything under oracle/"""
    pkg = ROOT / "hexl-fpga_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.cpp")) \
            + list(pkg.rglob("Makefile")):
        text = f.read_text()
        assert "liborc" not in text and "hexl_oracle" not in text and "import orc" not in text, f
    for so in (pkg / "lib").glob("*.so"):
        ldd = subprocess.run(["ldd", str(so)], capture_output=True, text=True).stdout
        assert "liborc" not in ldd
