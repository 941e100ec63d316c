import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, so when a test uses both
# torch and libcilqr_hip.so, torch has to be loaded first (as bench.py does).
try:  # pragma: no cover - depends on the box
    import torch  # noqa: F401
except Exception:
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the HIP library and the oracle once per session (both are in-tree .so files)."""
    import __graft_entry__ as ge
    ge.build()
    return True
