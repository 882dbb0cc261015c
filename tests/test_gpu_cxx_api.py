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
