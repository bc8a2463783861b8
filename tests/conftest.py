import os
import sys

import pytest

# torch ships its own HIP runtime (same SONAME as /opt/rocm's, which libmrt.so links): whichever is
# loaded first serves the whole process, and torch does not find the GPU on the other one. Tests
# that hand torch device buffers to the library therefore need torch imported before libmrt.so.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_ffi
    return oracle_ffi.load()
