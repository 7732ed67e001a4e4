import os
import sys

import pytest

# PyTorch ships its own libamdhip64.so.7 and libhfx.so links the system one with the same soname: whichever is mapped first
# serves both.  Tests that use torch next to libhfx in this process (device-buffer aliasing, the threaded 8-rank transport)
# need torch's copy to come first, as it does in bench.py and in the spawned workers -- so map it before any libhfx load.
try:
    import torch  # noqa: F401,E402
except ImportError:  # the oracle / host-mirror tests on the CPU do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "hifiles-solver_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.load()
