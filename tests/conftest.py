import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from jellyfish_amd import capi
        return capi.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """The HIP engine on a real device.  GPU tests FAIL (not skip) when the
    extension is missing on a GPU box: there is no fallback path to hide behind."""
    from jellyfish_amd import capi
    capi.load()
    n = capi.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests must run on the GPU box"
    return capi
