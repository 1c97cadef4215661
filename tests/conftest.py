import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ref_feeder():
    import oracle_lib
    try:
        return oracle_lib.RefFeeder()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref/libref_feeder.so not built (reference tree absent)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """A describe+match context on cuda:0 (640x480, maxkp 20000).  No fallback: fails loudly."""
    from coloc_amd import Context
    ctx = Context(device=0, width=640, height=480, maxkp=20000)
    yield ctx
    ctx.close()
