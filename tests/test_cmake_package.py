"""SURVEY 8f.3 -- the package config a downstream project consumes (`find_package(hexl-fpga)`, target hexl-fpga::hexl-fpga),
exercised the way the reference's consumers use theirs:

* tests/cmake_consumer/ is a downstream CMake project (the role of experimental/bridge-seal/tests/CMakeLists.txt:10,29-33) that
  builds the mini-CKKS flow example (examples/ckks_flow_example.cpp, standing in for bridge-seal's keyswitch-example.cpp:119-206:
  encrypt -> multiply -> relinearize -> rescale -> rotate -> decrypt, |error| < 5e-5, prime chain 52,30,30,40,27,27,27);
* where the reference tree exists, the REFERENCE's own examples/CMakeLists.txt (:9-24) is configured against the same package,
  unmodified, out of tree.

CPU: configure + build (the binaries land under tests/cmake_consumer/_build*, git-ignored, and travel to the GPU box), and the
CKKS flow itself runs against the CPU fake of the C-ABI (tests/cpp/fake_mi355x.cpp: oracle compute behind the REAL host layer).
GPU: the built binaries run on the MI355X."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "cmake_consumer"
BUILD, BUILD_REF = SRC / "_build", SRC / "_build_ref"
REF_EXAMPLES = Path("/root/reference/examples")


def cmake_build(src, build):
    assert shutil.which("cmake"), "cmake is part of the image"
    cfg = subprocess.run(["cmake", "-S", str(src), "-B", str(build), f"-DCMAKE_PREFIX_PATH={ROOT / 'cmake'}"],
                         capture_output=True, text=True)
    print(cfg.stdout[-1500:], cfg.stderr[-1500:])
    assert cfg.returncode == 0, "find_package(hexl-fpga) failed"
    bld = subprocess.run(["cmake", "--build", str(build)], capture_output=True, text=True)
    print(bld.stdout[-1500:], bld.stderr[-1500:])
    assert bld.returncode == 0
    return cfg.stdout


def test_downstream_project_finds_the_package_and_links():
    subprocess.run(["make", "-C", str(ROOT / "hexl-fpga_amd" / "host")], check=True, capture_output=True)
    cmake_build(SRC, BUILD)
    for exe in ("ckks_flow_example", "ckks_keyswitch_example"):
        out = subprocess.run(["readelf", "-d", str(BUILD / exe)], capture_output=True, text=True).stdout
        assert "libhexl-fpga.so" in out, "the imported target did not reach the link line"


@pytest.mark.skipif(not REF_EXAMPLES.exists(), reason="reference tree not present (GPU box)")
def test_reference_examples_cmakelists_configures_against_the_package():
    out = cmake_build(REF_EXAMPLES, BUILD_REF)
    assert "Intel HE Acceleration Library for FPGAs: found" in out          # examples/CMakeLists.txt:13-17
    assert (BUILD_REF / "example_dyadic_multiply").exists()


def test_ckks_flow_on_the_cpu_fake(orc):
    """the example's own logic (embedding, keys, rescale, Galois automorphism, CRT) pinned without a GPU: the same source
    against the real host layer over the oracle-backed fake C-ABI, N = 2048 and 4096"""
    orc.build()
    subprocess.run(["make", "-C", str(ROOT / "tests" / "cpp"), "host_tsan"], check=True, capture_output=True)
    host = ROOT / "tests" / "cpp" / "_host"
    exe = host / "ckks_flow_example"
    subprocess.run(["g++", "-O2", "-std=c++17", "-fsanitize=thread", f"-I{ROOT / 'include'}", "-o", str(exe),
                    str(ROOT / "examples" / "ckks_flow_example.cpp"), f"-L{host}", "-lhexl-fpga", f"-Wl,-rpath,{host}",
                    f"-Wl,-rpath,{ROOT / 'oracle'}", f"-Wl,-rpath-link,{host}"], check=True)
    for logn in ("11", "12"):
        out = subprocess.run([str(exe), logn, "1"], capture_output=True, text=True, timeout=600)
        print(out.stdout[-800:], out.stderr[-800:])
        assert out.returncode == 0 and "EXAMPLE PASSED" in out.stdout and "ThreadSanitizer" not in out.stderr


@pytest.mark.gpu
def test_ckks_flow_on_the_gpu():
    """experimental/bridge-seal/tests/seal_test.sh:20: N = 16384, 52,30,30,40,27,27,27, scale 2^52, three loops, 5e-5"""
    exe = BUILD / "ckks_flow_example"
    assert exe.exists(), "built by __graft_entry__.build() / the CPU test above"
    out = subprocess.run([str(exe), "14", "3"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "hexl-fpga_amd" / "lib")))
    print(out.stdout[-1500:], out.stderr[-1500:])
    assert out.returncode == 0 and out.stdout.count("SUCCESS") == 3 and "EXAMPLE PASSED" in out.stdout
    small = subprocess.run([str(exe), "12", "1"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "hexl-fpga_amd" / "lib")))
    assert small.returncode == 0 and "EXAMPLE PASSED" in small.stdout


@pytest.mark.gpu
def test_reference_example_built_by_its_own_cmakelists_runs():
    exe = BUILD_REF / "example_dyadic_multiply"
    if not exe.exists():
        pytest.skip("built only where the reference tree exists")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "hexl-fpga_amd" / "lib")))
    print(out.stdout[-1500:], out.stderr[-1500:])
    assert out.returncode == 0
