import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/liboracle.so), built on first use."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hiplib():
    """libtracs_hip.so, built in-tree if missing or stale (hipcc cross-compiles without a GPU)."""
    from tracs_amd import build as b
    b.build(force=False)
    from tracs_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
