import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nr-slam_amd", "py"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """CPU-side tests only need the shared library to exist; build it if it is missing."""
    import nrs
    if not os.path.exists(nrs.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return nrs


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the GPU tests; fails loudly (no CPU fallback) when there is no device."""
    import nrs
    c = nrs.Context()
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_exact():
    """Context with nrs_options.exact_trials = 1: every LM trial is solved to pcg_rtol."""
    import nrs
    c = nrs.Context(exact_trials=1)
    yield c
    c.close()
