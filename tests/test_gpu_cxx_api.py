"""GPU: the reference's public C++ API (libhexl-fpga.so, include/hexl-fpga.h) driven from C++ exactly as the
reference's gtests / benchmarks drive it; results checked bit-for-bit against the oracle inside the driver."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_cxx_api_driver():
    exe = ROOT / "tests" / "cpp" / "test_cxx_api"
    if not exe.exists():
        subprocess.run(["make", "-C", str(exe.parent)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ALL PASSED" in out.stdout


def test_cxx_api_driver_tiny_staging_slabs():
    """same driver with 1 MiB staging slabs: every batch becomes many sub-batches with a ragged tail, exercising the
    double-buffered H2D / kernel / D2H pipeline of the host-pointer entry points (capi.hip run_pipeline)"""
    import os
    exe = ROOT / "tests" / "cpp" / "test_cxx_api"
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HEXL_HOST_SUB_MB="1", HEXL_HOST_THREADS="3"))
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ALL PASSED" in out.stdout


def test_ckks_keyswitch_example():
    """examples/ckks_keyswitch_example.cpp: real RLWE switching keys, _NTT/_INTT + KeySwitch through the public
    API; decryption under the old key recovers t*s_new up to small noise (what the reference's SEAL test checks)"""
    exe = ROOT / "examples" / "ckks_keyswitch_example"
    if not exe.exists():
        subprocess.run(["make", "-C", str(exe.parent)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    print(out.stdout[-2000:], out.stderr[-1000:])
    assert out.returncode == 0 and "EXAMPLE PASSED" in out.stdout
