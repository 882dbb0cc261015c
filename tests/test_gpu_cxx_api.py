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


def test_aliased_results_across_staging_sub_batches():
    """benchmark/bench_keyswitch.cpp:113-131: 40 iterations x 2 vectors in one worksize window, all accumulating into the
    same two result arrays, with 1 MiB staging slabs (every object its own sub-batch), repeated 50 times: the
    accumulating unpacks of consecutive sub-batches must run in submission order"""
    import os
    exe = ROOT / "tests" / "cpp" / "test_cxx_api"
    out = subprocess.run([str(exe), "alias", "50"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HEXL_HOST_SUB_MB="1", HEXL_HOST_THREADS="4"))
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ALL PASSED" in out.stdout


@pytest.mark.parametrize("env", [{}, {"NUM_DEV": "2", "HEXL_DEV_ALIAS": "1", "HEXL_HOST_SUB_MB": "8"}], ids=["one_device", "two_runners_small_slabs"])
def test_randomised_windows_every_object_checked(env):
    """round 6: random primitive / ring dimension / modulus size / worksize (up to hundreds of objects: persistent kernels, several staging
    sub-batches, many small workgroups per CU) through the host-pointer API, EVERY object of every window against the oracle"""
    import os
    exe = ROOT / "tests" / "cpp" / "test_cxx_api"
    out = subprocess.run([str(exe), "stress", "20", "7"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ALL PASSED" in out.stdout


def test_two_device_contexts_on_one_gpu():
    """NUM_DEV=2 with HEXL_DEV_ALIAS=1: two runner threads, two contexts and two plan caches on the one visible GPU --
    the in-process multi-device path (DevicePool, host/src/fpga.cpp:1646-1673) executes for real, including the
    aliased-result ordering between devices"""
    import os
    exe = ROOT / "tests" / "cpp" / "test_cxx_api"
    env = dict(os.environ, NUM_DEV="2", HEXL_DEV_ALIAS="1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ALL PASSED" in out.stdout and out.stdout.count("hexl_mi355x: device") == 2
    out = subprocess.run([str(exe), "alias", "10"], capture_output=True, text=True, timeout=900,
                         env=dict(env, HEXL_HOST_SUB_MB="1"))
    print(out.stdout[-2000:], out.stderr[-2000:])
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
