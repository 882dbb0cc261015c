"""GPU: the REFERENCE's own gtest and google-benchmark sources, compiled unmodified against libhexl-fpga.so
(tests/ref_harness/Makefile; gtest/benchmark API shims), run against the MI355X kernels. The binaries exist only
where the reference tree was available at build time (like oracle/_ref); skipped otherwise."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
BUILD = Path(__file__).resolve().parent / "ref_harness" / "_build"


@pytest.mark.parametrize("exe", ["test_fwd_ntt", "test_inv_ntt", "test_dyadic_multiply"])
def test_reference_gtest_binary(exe):
    path = BUILD / exe
    if not path.exists():
        pytest.skip("reference test sources were not built on this box")
    out = subprocess.run([str(path)], capture_output=True, text=True, timeout=1200)
    tail = out.stdout[-1500:]
    print(tail, out.stderr[-500:])
    assert out.returncode == 0 and "[  PASSED  ]" in out.stdout, tail


@pytest.mark.parametrize("exe", ["bench_fwd_ntt", "bench_inv_ntt", "bench_dyadic_multiply"])
def test_reference_benchmark_binary_runs(exe):
    path = BUILD / exe
    if not path.exists():
        pytest.skip("reference benchmark sources were not built on this box")
    out = subprocess.run([str(path), "--benchmark_min_time=0.05"], capture_output=True, text=True, timeout=1200)
    print(out.stdout[-1500:], out.stderr[-500:])
    assert out.returncode == 0 and "ms" in out.stdout
