import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _no_internal_errors():
    """a green suite must mean that no merge loop stopped on its own consistency check.  The library never runs a
    loop twice (a violated invariant is GLIA_HMT_ERR_INTERNAL); this asserts that not one call of the whole session ended that
    way, whatever the tests themselves looked at."""
    yield
    from glia_amd import hmt
    if hmt._lib is not None:
        n = hmt.Context.internal_errors()
        assert n == 0, "%d call(s) of this session ended with GLIA_HMT_ERR_INTERNAL (glia_hmt_internal_errors)" % n
