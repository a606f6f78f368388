import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def native_lib():
    """Build (if needed) and load libdf3d_hip.so; hipcc cross-compiles without a GPU."""
    from deepfly3d_amd import _native, build

    if not os.path.exists(_native.LIB_PATH):
        build.build()
    return _native.load()


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False: these tests never fall back to the CPU")
    return torch.device("cuda:0")
