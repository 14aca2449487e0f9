import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_present():
    try:
        from deconv3d_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a device must FAIL loudly, not skip silently:
    # only auto-skip gpu tests when the user did not ask for them.
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
