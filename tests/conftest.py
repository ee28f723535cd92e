import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'sub-cortical_segmentation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, 'tests', 'golden')
WEIGHTS = os.path.join(PKG, 'nets', 'miccai2012_v1', 'miccai2012_v1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def exact_f32_form(obj):
    """Pin a libscs handle (or a Net's) to the exact-f32 kernels whatever SCS_FWD_SPLIT / SCS_TRAIN_SPLIT say in the environment: for the
    tests that compare two FORMS of the f32 path bit for bit (the split forms give the oracle's numbers to the same bars, not the f32
    path's bits: tests/test_gpu_split.py, test_gpu_train_split.py).  Returns the object."""
    h = obj._ensure_handle() if hasattr(obj, '_ensure_handle') else obj
    h.set_option('fwd_split', 0)
    h.set_option('train_split', 0)
    return obj


@pytest.fixture(scope='session')
def weights():
    from cnn_cort import params as P
    return P.load_params(WEIGHTS)


@pytest.fixture(scope='session')
def handle(weights):
    """libscs handle on cuda:0 with the shipped weights (GPU tests only). Fails loudly if the .so is missing."""
    import torch
    from cnn_cort import _hip, params as P
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    h = _hip.Handle(0)
    h.set_params(P.params_to_flat(weights))
    yield h
    h.close()
