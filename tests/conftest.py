import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs the HIP path through the C ABI)")


@pytest.fixture(scope="session")
def built():
    """Build the product library, the oracle and the host-compiled device mirror once per session."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built):
    from glome_amd import api
    ctx = api.Context(0)  # raises (no fallback) if no MI355X is usable
    yield ctx
    ctx.close()
