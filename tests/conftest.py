import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own HIP runtime; libspgemm_hip.so links the system one.  Both live happily in one
    process only if torch brings its runtime up FIRST (measured on the GPU box: library first -> torch reports "No HIP
    GPUs are available"; torch first -> both work).  Tests that use dist.HipEngine (torch tensors + the library) rely on
    this order, so on a GPU box torch's context is created before any test touches the library."""
    try:
        import torch
        if torch.cuda.device_count() > 0 and torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """Compile oracle/liboracle.so (and _ref when /root/reference exists) once per session."""
    from oracle import pyoracle as po
    po.build(ref=True)
    yield
