import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native libraries, built once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built):
    from lupinpathtracer_amd import api
    if api.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests must run on the GPU box (the product has no CPU fallback)")
    ctx = api.Context(0)
    yield ctx
    ctx.close()
