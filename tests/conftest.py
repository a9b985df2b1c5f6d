import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the restated dataset arithmetic mixes NumPy scalars with torch tensors exactly like the reference does; NumPy 2 warns
    # about torch's __array_wrap__ signature there
    config.addinivalue_line("filterwarnings", "ignore:__array_wrap__ must accept context:DeprecationWarning")


@pytest.fixture(scope="session")
def built_lib():
    """Path of the in-tree HIP library (built on demand; hipcc cross-compiles without a GPU)."""
    from sonicsim_amd import build
    return build.build()


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no ROCm device is visible (the HIP path has no CPU fallback)")
    from sonicsim_amd import ops
    ops.init(0)
    return torch.device("cuda:0")

