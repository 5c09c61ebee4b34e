import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import helpers
    return helpers.load_oracle()


@pytest.fixture(scope="session")
def hip():
    """The HIP library initialised on cuda:0.  Fails (not skips) when missing."""
    import torch
    from uvg266_amd import lib
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    return lib.init(0)
