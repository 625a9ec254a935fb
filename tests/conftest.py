import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes over libptcore.so).  Built on demand; no fallback if that fails."""
    if not os.path.exists(os.path.join(graft.PKG_DIR, "libptcore.so")):
        graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle binding (tests/oracle_api.py) -- the checker, never the thing under test."""
    return graft.load_oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
