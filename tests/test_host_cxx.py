"""CPU: the REAL C++ host layer (hexl-fpga_amd/host/hexl_fpga_api.cpp: submission FIFO, one runner thread per device,
fences, NUM_DEV sharding, output-aliasing rule, condition-variable completion) built with ThreadSanitizer against a CPU
fake of the C-ABI (tests/cpp/fake_mi355x.cpp, oracle compute) and driven by the same C++ driver the GPU box runs
(tests/cpp/test_cxx_api.cpp). The fake aborts if two threads ever enter one device context at the same time."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "tests" / "cpp" / "_host" / "test_cxx_api"


@pytest.fixture(scope="module")
def exe(orc):
    orc.build()
    subprocess.run(["make", "-C", str(EXE.parent.parent), "host_tsan"], check=True, capture_output=True)
    return EXE


def run(exe, args, **env):
    out = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **{k: str(v) for k, v in env.items()}))
    print(out.stdout[-2000:], out.stderr[-3000:])
    assert "ThreadSanitizer" not in out.stderr, "data race in the host layer"
    assert out.returncode == 0 and "ALL PASSED" in out.stdout
    return out


@pytest.mark.parametrize("num_dev", [1, 3])
def test_driver_on_fake_devices_under_tsan(exe, num_dev):
    run(exe, [], NUM_DEV=num_dev, FAKE_DELAY_US=200)


def test_aliased_results_are_accumulated_in_order_across_devices(exe):
    """benchmark/bench_keyswitch.cpp:113-131 shape: 40 x 2 objects in one window, two result arrays; with three
    runner threads the output-aliasing rule has to serialise them"""
    run(exe, ["alias", "5"], NUM_DEV=3, FAKE_DELAY_US=50)


def test_debug_trace_and_small_runs(exe):
    """FPGA_DEBUG=1 prints one line per run; FPGA_BUFSIZE=1 forces one object per run (host/src/fpga_int.cpp:123-129)"""
    out = run(exe, ["alias", "1"], NUM_DEV=2, FPGA_DEBUG=1, FPGA_BUFSIZE=1)
    assert "x KeySwitch" in out.stderr


def test_concurrent_worksize1_callers_wait_for_their_own_object(exe):
    """with NUM_DEV > 1 and several submitting threads a worksize-1 call must not return before ITS object is
    done. Device 0 is 8x slower here, so later objects on the other devices finish first -- a completion counter would
    release the caller early; tickets retire in order."""
    run(exe, ["threads", "4", "15"], NUM_DEV=3, FAKE_DELAY_US=300, FAKE_DELAY_DEV0_X=8)


@pytest.mark.parametrize("num_dev", [1, 3])
def test_randomised_windows_under_tsan(exe, num_dev):
    """round 6: the driver's stress mode (random primitive / ring dimension / worksize up to hundreds of objects, every object against
    the oracle) on the real host layer: FIFO, runs of compatible objects, sharding over the runners, completion -- under ThreadSanitizer"""
    run(exe, ["stress", "8", str(num_dev)], NUM_DEV=num_dev)
