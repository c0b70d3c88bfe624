import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    import sregex_amd as S
    if not os.path.exists(S.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return S.load_library()


@pytest.fixture(scope="session")
def blocks():
    import harness
    return harness.load_blocks()
