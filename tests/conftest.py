import os
import sys

import pytest

# a mailbox take that never gets its flag should fail a test in seconds, not after the production bound of 300 s
os.environ.setdefault("PERCNN_PEER_TIMEOUT_S", "20")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A source-only tree (no in-tree libpercnn_pi.so yet) is compiled once before the first test; the package itself
    never builds on import and never falls back."""
    from percnn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        _lib.build()


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")
