import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "scs-python_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _have_gpu():
    try:
        from scs import _scs_hip
        return _scs_hip.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a usable GPU must FAIL, not pass on skips: every gpu-marked test gets a
    setup-time check (tests selected without the marker expression are left alone)."""
    if "gpu" not in (config.getoption("-m") or "") or "not gpu" in (config.getoption("-m") or ""):
        return
    if _have_gpu():
        return
    for item in items:
        if item.get_closest_marker("gpu"):
            item.add_marker(pytest.mark.usefixtures("_require_gpu"))


@pytest.fixture
def _require_gpu():
    if not _have_gpu():
        pytest.fail("-m gpu was requested but no HIP device / libscs_hip.so is usable (no CPU fallback exists)")


@pytest.fixture(scope="session")
def gpu_available():
    return _have_gpu()
