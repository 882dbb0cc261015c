import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc   # oracle/orc.py (test infrastructure)
    _orc.build(with_ref=True)
    return _orc


@pytest.fixture(scope="session")
def hx():
    import hexl_fpga_amd
    return hexl_fpga_amd


@pytest.fixture(scope="session")
def ctx(hx):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    c = hx.Context(0)
    yield c
    c.close()


@pytest.fixture
def dev():
    import torch
    return torch.device("cuda:0")


def stimulus(kind: str, n: int, q: int, seed: int = 7) -> np.ndarray:
    """the reference's test stimuli (tests/test_fwd_ntt.cpp:60-101)"""
    if kind == "RANDOM":
        rng = np.random.default_rng(seed)
        return rng.integers(0, q, size=n, dtype=np.uint64)
    if kind == "RAMP":
        return np.arange(n, dtype=np.uint64)
    if kind == "ALL_ZEROS":
        return np.zeros(n, dtype=np.uint64)
    if kind == "ALL_ONES":
        return np.ones(n, dtype=np.uint64)
    if kind == "IMPULSE":
        x = np.zeros(n, dtype=np.uint64)
        x[0] = 1
        return x
    if kind == "ALL_MAX_VALUES":
        return np.full(n, 2**64 - 1, dtype=np.uint64)
    raise KeyError(kind)
