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


def unshuffle_f32(planes, shape):
    """Inverse of generate_golden.shuffle_f32: byte planes -> f32 raster."""
    return np.ascontiguousarray(planes.T).view('<f4').reshape(shape)


def load_g10(name='g10_10m.npz'):
    """G10 (the 10 m regime) / G11 (wandering tracks) with their two rasters unpacked."""
    g = dict(load_golden(name))
    shape = tuple(int(x) for x in g['shape'])
    g['shape'] = shape
    g['orograph_f32'] = unshuffle_f32(g.pop('orograph_f32_planes'), shape)
    g['potential'] = unshuffle_f32(g.pop('potential_planes'), shape)
    hist = np.zeros(shape, dtype=np.int32)
    hist[g['hist_rows'].astype(int), g['hist_cols'].astype(int)] = g['hist_vals']
    g['hist'] = hist
    return g


@pytest.fixture(scope='session')
def g10():
    return load_g10()


@pytest.fixture(scope='session')
def g11():
    return load_g10('g11_wander.npz')


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
