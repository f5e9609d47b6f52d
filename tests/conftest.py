import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(scope='session')
def gpu():
    """Skip-free guard: -m gpu tests must run on a GPU box with the HIP library."""
    import torch
    assert torch.cuda.is_available(), 'gpu-marked test started without a GPU'
    from ssrs_amd import _native
    _native.lib()      # raises if libssrs_hip.so is missing
    return torch.device('cuda', 0)
