import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_available() -> bool:
    try:
        import ctypes
        from qwen3_tts_axera_russian_amd import LIB_PATH
        return ctypes.CDLL(LIB_PATH).q3_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP library on a box with a GPU; fails loudly (never skips to a fallback) when the
    extension is missing."""
    from qwen3_tts_axera_russian_amd import hiplib
    lib = hiplib.load()
    if lib.q3_device_count() <= 0:
        pytest.fail("test marked gpu but no HIP device is visible")
    return lib


@pytest.fixture(scope="session")
def test_lib(gpu_lib):
    """Kernel-level hooks (lib/libqwen3tts_test.so): tests only, never the product path."""
    from qwen3_tts_axera_russian_amd import hiplib
    return hiplib.load_test()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "frontend_golden.npz"))
