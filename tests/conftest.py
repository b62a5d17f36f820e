import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle (oracle/smc_oracle.c), built on demand. Test infrastructure only."""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def L():
    """ctypes binding of the product library libsmchip.so (must already be built)."""
    from sequential_monte_carlo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    _lib.lib()
    return _lib


GOLDEN = os.path.join(ROOT, "tests", "golden")
