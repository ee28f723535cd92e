"""Consumes outputs of the UNMODIFIED reference (python2 + Theano / Lasagne / nolearn) on two committed fixtures, when somebody with
such a machine has produced them (`tests/golden/ref_g1_random.npz`, `ref_g3_classes.npz`, arrays `probs`
[N, 15] float32 and `labels` [N]).  Until then every test here SKIPS, and parity stays "unpinned by the reference" (DESIGN.md 3).
With the files present: the float64 oracle against the reference (CPU), and the HIP path against the reference at BASELINE.json's bar -
probabilities within 1e-4, labels bit-exact (`-m gpu`)."""
import os

import numpy as np
import pytest

from conftest import GOLD, gold

NAMES = ['g1_random', 'g3_classes']


def ref(name):
    path = os.path.join(GOLD, 'ref_%s.npz' % name)
    if not os.path.exists(path):
        pytest.skip('no reference outputs for %s (tests/golden/ref_%s.npz): see INTEGRATION.md section 6' % (name, name))
    r = np.load(path)
    g = gold(name + '.npz')
    n = g['ax'].shape[0]
    assert r['probs'].shape == (n, 15) and r['labels'].shape == (n,), 'ref_%s.npz does not belong to %s.npz' % (name, name)
    assert np.abs(r['probs'].sum(1) - 1).max() < 1e-4 and np.array_equal(r['labels'], r['probs'].argmax(1))
    return g, np.asarray(r['probs'], np.float64), np.asarray(r['labels'], np.int64)


@pytest.mark.parametrize('name', NAMES)
def test_oracle_against_the_reference(weights, name):
    from oracle import np_oracle as O
    g, p_ref, l_ref = ref(name)
    p = O.forward(weights, g['ax'], g['co'], g['sa'], g['atlas'])
    assert np.abs(p - p_ref).max() < 1e-5, 'the float64 restatement is %g from the reference' % np.abs(p - p_ref).max()
    assert np.array_equal(p.argmax(1), l_ref)


@pytest.mark.gpu
@pytest.mark.parametrize('split', [False, True])
@pytest.mark.parametrize('name', NAMES)
def test_hip_path_against_the_reference(handle, name, split):
    from test_gpu_forward import run_forward
    g, p_ref, l_ref = ref(name)
    handle.set_option('fwd_split', 1 if split else -1)
    try:
        probs, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
    finally:
        handle.set_option('fwd_split', -1)
    assert np.abs(probs - p_ref).max() < 1e-4
    assert np.array_equal(labels, l_ref)
