import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ffp():
    import ffp_amd  # noqa: F401  (alias of the hyphenated package directory)
    return ffp_amd


@pytest.fixture(scope="session")
def gpu_lib(ffp):
    """The loaded HIP library on a machine with a GPU. Fails (not skips) when the library is missing: GPU tests
    must exercise the native path."""
    # torch carries its own copy of the HIP runtime and must be brought up before libffp.so initialises the system copy
    # (ffp_amd._lib.lib() does this by itself when torch is already imported)
    import torch  # noqa: F401
    from ffp_amd import _lib
    l = _lib.lib()
    assert _lib.device_count() > 0, "no HIP device visible"
    return _lib
