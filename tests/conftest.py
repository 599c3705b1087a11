import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def lib():
    """The C-ABI HIP library; building it is part of __graft_entry__.build()."""
    from inverse_audio_synthesis_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    return torch.device("cuda:0")
