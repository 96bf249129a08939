import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def _have_gpu():
    try:
        from toppra_amd import _capi
        return _capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """Initialise the HIP library; GPU tests must fail loudly (not skip) without the extension."""
    from toppra_amd import _capi
    _capi.init(0)
    return _capi


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def reference():
    """The real reference package (build container only)."""
    from oracle import ref_loader
    mod = ref_loader.load()
    if mod is None:
        pytest.skip("reference not available on this machine")
    return mod
