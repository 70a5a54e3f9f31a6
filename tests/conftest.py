import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "wildcat-slam_amd", "python"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu():
    """A wc_ctx on cuda:0.  Fails loudly (no CPU fallback) when the HIP library or the device is missing."""
    from wildcat_slam_amd import lib

    ctx = lib.Context(0)
    yield ctx
    ctx.close()
