"""CPU: the exact FP64 modular arithmetic the keyswitch kernels use on gfx950 (hexl-fpga_amd/csrc/f64_arith.hpp)
is plain IEEE-754 double mul/add/fma/rint, so the very same source is validated on the host against exact
128-bit integer arithmetic and the oracle's canonical transforms (tests/cpp/f64_selftest.cpp)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_f64_arith_selftest(orc):
    d = ROOT / "tests" / "cpp"
    subprocess.run(["make", "-C", str(d), "f64_selftest"], check=True)
    out = subprocess.run([str(d / "f64_selftest")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:])
    assert out.returncode == 0 and "ALL PASSED" in out.stdout
