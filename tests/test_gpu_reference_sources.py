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


def test_reference_example_program():
    """examples/examples.cpp + example_dyadic_multiply.cpp unmodified (the reference's `make examples` target,
    examples/CMakeLists.txt:18-24): 40 multiplications at N = 8192, 6 moduli, checked against the example's own CPU result"""
    path = BUILD / "example_dyadic_multiply"
    if not path.exists():
        pytest.skip("reference example sources were not built on this box")
    out = subprocess.run([str(path)], capture_output=True, text=True, timeout=600)
    print(out.stdout[-1500:], out.stderr[-500:])
    assert out.returncode == 0 and "Correct multiplication: both vectors are equal" in out.stdout
    assert "Error" not in out.stdout


def _vectors(tmp, n, shapes, count=2):
    """the directory the reference sources read their JSON vectors from. If the caller's environment already points
    KEYSWITCH_DATA_DIR at the OFFICIAL vectors (testdata.zip, README.md:166-176: files <n>_<L>_<K>_<rns>_2_*.json), those are
    used as they are -- the only reference-held ground truth for the keyswitch; otherwise vectors in the same format are
    generated from this repository's oracle (tests/ref_harness/make_ks_vectors.py)."""
    import glob
    import os
    import sys
    official = os.environ.get("KEYSWITCH_DATA_DIR")
    if official and all(glob.glob(os.path.join(official, f"{n}_{L}_{K}_{rns}_2_*.json")) for (L, K, rns) in shapes):
        print(f"using the official keyswitch vectors in {official}")
        return official
    sys.path.insert(0, str(BUILD.parent))
    import make_ks_vectors
    for (L, K, rns) in shapes:
        make_ks_vectors.main(str(tmp), n, L, K, rns, count)
    return str(tmp)


def test_reference_keyswitch_gtest(tmp_path):
    """tests/test_keyswitch.cpp unmodified: JSON loader -> KeySwitch(worksize batch, caller twiddles) -> ASSERT_EQ"""
    exe = BUILD / "test_keyswitch"
    if not exe.exists():
        pytest.skip("reference keyswitch test was not built on this box")
    import os
    data = _vectors(tmp_path, 4096, [(6, 7, 7), (5, 7, 6)])
    env = dict(os.environ, KEYSWITCH_DATA_DIR=data, N="4096")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=1200, env=env)
    print(out.stdout[-1500:], out.stderr[-500:])
    assert out.returncode == 0 and "[  PASSED  ] 2 test(s)" in out.stdout


def test_reference_dyadic_keyswitch_gtest(tmp_path):
    """tests/test_dyadic_multiply_keyswitch.cpp unmodified (N = 16384 vectors, both primitives interleaved)"""
    exe = BUILD / "test_dyadic_multiply_keyswitch"
    if not exe.exists():
        pytest.skip("reference combined test was not built on this box")
    import os
    data = _vectors(tmp_path, 16384, [(6, 7, 7)], count=2)
    env = dict(os.environ, KEYSWITCH_DATA_DIR=data)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=1800, env=env)
    print(out.stdout[-1500:], out.stderr[-500:])
    assert out.returncode == 0 and "[  PASSED  ]" in out.stdout


def test_reference_keyswitch_benchmark(tmp_path):
    """benchmark/bench_keyswitch.cpp unmodified: ITER x the 16384_6_7_7_2 vectors in ONE worksize window, the same
    result arrays submitted repeatedly (accumulation order matters, see hexl_keyswitch_host)"""
    exe = BUILD / "bench_keyswitch"
    if not exe.exists():
        pytest.skip("reference keyswitch benchmark was not built on this box")
    import os
    data = _vectors(tmp_path, 16384, [(6, 7, 7)], count=2)
    env = dict(os.environ, KEYSWITCH_DATA_DIR=data, ITER="4")
    out = subprocess.run([str(exe), "--benchmark_min_time=0.05"], capture_output=True, text=True, timeout=1800, env=env)
    print(out.stdout[-1500:], out.stderr[-500:])
    assert out.returncode == 0 and "16384_6_7_7_2" in out.stdout


# The reference's tests/micro_*.sh and benchmark/micro_*.sh run the same binaries under a matrix of environment
# settings (FPGA_BITSTREAM / FPGA_KERNEL select a bitstream, BATCH_SIZE_* the FPGA-side batching, N the keyswitch vector
# size, RUN_CHOICE the backend). This library has one backend and batches by itself; the variables must be accepted
# and must not change any result.
#
# The scripts themselves are reference sources: they cannot travel to the GPU box. So they are RUN where they lie, unmodified,
# as scripts, in the build container against recording stubs (tests/ref_harness/trace_scripts.py, RUN_CHOICE=1 as the reference
# documents for boxes without `aocl`), and what they executed -- binary, arguments, environment per invocation -- is replayed
# here against the real binaries on the MI355X, script by script, in the scripts' own order.
def _script_trace():
    import json
    f = BUILD / "script_trace.json"
    return json.loads(f.read_text()) if f.exists() else {"scripts": [], "invocations": []}


_KS_DIRS = {}


def _ks_data_dir(tmp_factory, n):
    """JSON vectors for the keyswitch binaries at ring dimension n, generated once per session"""
    if n not in _KS_DIRS:
        _KS_DIRS[n] = _vectors(tmp_factory.mktemp(f"ksvec{n}"), n, [(6, 7, 7), (5, 7, 6)], count=2)
    return _KS_DIRS[n]


@pytest.mark.parametrize("script", _script_trace()["scripts"] or ["<no trace: reference tree absent at build time>"])
def test_reference_script_replay(tmp_path_factory, script):
    """every invocation one of the reference's runner scripts made, replayed with the environment the script gave it"""
    import os
    runs = [e for e in _script_trace()["invocations"] if e["script"] == script]
    if not runs:
        pytest.skip("tests/ref_harness/_build/script_trace.json was not produced on this box")
    for e in runs:
        path = BUILD / e["exe"]
        assert path.exists(), f"{script} calls {e['exe']}, which was not built"
        env = dict(os.environ, RUN_CHOICE="1", **e["env"])
        if "keyswitch" in e["exe"]:
            env["KEYSWITCH_DATA_DIR"] = _ks_data_dir(tmp_path_factory, int(e["env"].get("N", "16384")))
        out = subprocess.run([str(path), *e["argv"]], capture_output=True, text=True, timeout=1800, env=env)
        what = f"{script}: {' '.join(f'{k}={v}' for k, v in e['env'].items())} ./{e['exe']}"
        print(what, "\n", out.stdout[-600:], out.stderr[-300:])
        assert out.returncode == 0, what
        if e["exe"].startswith("test_"):
            assert "[  PASSED  ]" in out.stdout and "FAILED" not in out.stdout, what
        else:
            assert " ms " in out.stdout, what


# The same matrix written out by hand (rounds 1-3; kept: it also runs where no trace was produced).
ENV_MATRIX = [
    ("test_fwd_ntt", {"FPGA_KERNEL": "NTT"}),
    ("test_fwd_ntt", {"FPGA_KERNEL": "NTT", "BATCH_SIZE_NTT": "8"}),
    ("test_inv_ntt", {"FPGA_KERNEL": "INTT", "BATCH_SIZE_INTT": "8"}),
    ("test_dyadic_multiply", {"FPGA_KERNEL": "DYADIC_MULTIPLY", "BATCH_SIZE_DYADIC_MULTIPLY": "8"}),
    ("test_keyswitch", {"FPGA_KERNEL": "KEYSWITCH", "N": "16384", "BATCH_SIZE_KEYSWITCH": "2"}),
    ("test_keyswitch", {"FPGA_KERNEL": "KEYSWITCH", "N": "8192", "BATCH_SIZE_KEYSWITCH": "1"}),
    ("test_dyadic_multiply_keyswitch", {"FPGA_KERNEL": "DYADIC_MULTIPLY_KEYSWITCH", "BATCH_SIZE_DYADIC_MULTIPLY": "2",
                                        "BATCH_SIZE_KEYSWITCH": "2"}),
]


@pytest.mark.parametrize("exe,extra", ENV_MATRIX, ids=[f"{e}-{'-'.join(f'{k}={v}' for k, v in x.items() if k != 'FPGA_KERNEL')}" for e, x in ENV_MATRIX])
def test_reference_env_matrix(tmp_path, exe, extra):
    path = BUILD / exe
    if not path.exists():
        pytest.skip("reference test sources were not built on this box")
    import os
    env = dict(os.environ, RUN_CHOICE="2", FPGA_BITSTREAM="/nonexistent/libkernel.so", **extra)
    if "keyswitch" in exe:
        n = int(extra.get("N", "16384"))
        env["KEYSWITCH_DATA_DIR"] = _vectors(tmp_path, n, [(6, 7, 7)] if "dyadic" in exe else [(6, 7, 7), (5, 7, 6)], count=2)
    out = subprocess.run([str(path)], capture_output=True, text=True, timeout=1800, env=env)
    print(out.stdout[-1200:], out.stderr[-400:])
    assert out.returncode == 0 and "[  PASSED  ]" in out.stdout
