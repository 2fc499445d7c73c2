import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library, initialised on device 0 (GPU tests only)."""
    import torch
    from countr_amd import _lib
    assert torch.cuda.is_available(), "GPU test running without a GPU"
    L = _lib.lib()
    _lib.check(L.countr_init(0), "countr_init")
    return L
