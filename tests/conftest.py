import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rtx():
    """The product library; GPU tests fail loudly if it is missing or no device is visible."""
    import raytracer_amd
    from raytracer_amd import _ffi
    lib = _ffi.load_library()
    assert lib.rtx_device_count() > 0, "no HIP device visible: GPU tests need a real MI355X"
    return raytracer_amd
