"""The split inference path (option fwd_split = 1: csrc/conv_split.h - convolutions on the f16 matrix cores, every f32 operand
carried as two f16 pieces, f32 accumulation) against the float64 oracle AT THE SAME BARS as the exact-f32 path
(tests/test_gpu_forward.py): probabilities within 1e-4, logits and per-layer activations within 2e-5 of their scale, labels exact
outside the near-tie band - and against the f32 path itself.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from conftest import gold
from oracle import np_oracle as O
from test_gpu_forward import PROB_TOL, check_labels, run_forward

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split(handle):
    handle.set_option('fwd_split', 1)
    handle.set_option('split_fallbacks', 0)
    yield handle
    handle.set_option('fwd_split', -1)


@pytest.mark.parametrize('name', ['g1_random.npz', 'g2_volume.npz', 'g3_classes.npz', 'g4_ties.npz'])
def test_golden_sets_on_the_split_path(handle, name):
    g = gold(name)
    n = g['ax'].shape[0]
    p32, l32 = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
    lg32 = handle.get_activation(7, n * 16).reshape(n, 16)[:, :15]
    handle.set_option('fwd_split', 1)
    handle.set_option('split_fallbacks', 0)
    try:
        probs, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
        logits = handle.get_activation(7, n * 16).reshape(n, 16)[:, :15]
        assert handle.get_option('split_fallbacks') == 0        # the result below IS the split kernels'
    finally:
        handle.set_option('fwd_split', -1)
    assert np.isfinite(probs).all()
    err = np.abs(probs - g['probs']).max()
    assert err < PROB_TOL, 'max |p - p_oracle| = %g' % err
    assert np.abs(probs.sum(1) - 1).max() < 1e-5
    check_labels(labels, g['probs'])
    assert np.array_equal(labels, probs.argmax(1))
    scale = max(1.0, float(np.abs(g['logits']).max()))
    lerr = float(np.abs(logits - g['logits']).max())
    lerr32 = float(np.abs(lg32 - g['logits']).max())
    assert lerr <= 2e-5 * scale, 'max |logit - logit_oracle| = %g (scale %g)' % (lerr, scale)
    print('%s: logit error vs the float64 oracle: split %.3g, exact-f32 path %.3g (scale %.3g); split vs f32 path: probabilities %.3g, '
          'labels differing %d' % (name, lerr, lerr32, scale, np.abs(probs - p32).max(), int((labels != l32).sum())))
    # the two device paths agree far inside the oracle bar
    assert np.abs(probs - p32).max() < 2e-5
    srt = np.sort(g['probs'], axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert np.array_equal(labels[clear], l32[clear])


def test_per_layer_activations_on_the_split_path(split, weights):
    g = gold('g2_volume.npz')
    n = 24
    args = [g[k][:n] for k in ('ax', 'co', 'sa', 'atlas')]
    tr = {}
    O.forward(weights, *args, trace=tr)
    run_forward(split, *args)
    towers = ('axial', 'coronal', 'saggital')

    def close(got, ref, what):
        scale = max(1.0, np.abs(ref).max())
        err = np.abs(got - ref).max()
        assert err <= 2e-5 * scale, '%s: max err %g (scale %g)' % (what, err, scale)

    for which, key, per in ((0, 'pool1', 20 * 196), (1, 'conv3', 40 * 144), (2, 'pool2', 40 * 25), (3, 'conv5', 540)):
        got = split.get_activation(which, 3 * n * per).reshape(3, n, -1)
        for t, tn in enumerate(towers):
            close(got[t], tr['%s_%s' % (tn, key)].reshape(n, -1), '%s_%s' % (tn, key))
    # the dense head on the split kernels (csrc/dense_split.h): rows of halves between the layers, handed back in the f32 layouts
    feat = split.get_activation(4, n * 540).reshape(n, 540)
    close(feat, np.concatenate([tr['%s_d1' % t] for t in towers], axis=1), 'd1 concat')
    fc1 = split.get_activation(5, n * 556).reshape(n, 556)
    close(fc1[:, :540], tr['FC1'], 'FC1')
    assert np.abs(fc1[:, 540:555] - args[3]).max() <= 2e-7 and np.all(fc1[:, 555] == 0)     # the prior: two f16 pieces of each value
    fc2 = split.get_activation(6, n * 272).reshape(n, 272)
    close(fc2[:, :270], tr['fc_2'], 'fc_2')
    logits = split.get_activation(7, n * 16).reshape(n, 16)
    close(logits[:, :15], tr['logits'], 'logits')
    assert split.get_option('split_fallbacks') == 0
    # ... and with the head left on the f32-MFMA GEMMs (option split_head = 0: the towers alone on the split kernels)
    split.set_option('split_head', 0)
    try:
        run_forward(split, *args)
        close(split.get_activation(4, n * 540).reshape(n, 540), feat, 'd1 concat, f32 head vs split head')
        close(split.get_activation(7, n * 16).reshape(n, 16), logits, 'logits, f32 head vs split head')
    finally:
        split.set_option('split_head', -1)


@pytest.mark.parametrize('n', [1, 3, 17, 700, 4099])
def test_ragged_batches_on_the_split_path(handle, n):
    """Persistent workgroups, 4 / 2 / 4 patch-towers per iteration in conv3 / conv4 / conv5: every remainder."""
    rng = np.random.default_rng(n)
    ax, co, sa = (rng.standard_normal((n, 1, 32, 32)).astype(np.float32) for _ in range(3))
    at = rng.dirichlet(np.ones(15), n).astype(np.float32)
    p32, l32 = run_forward(handle, ax, co, sa, at)
    handle.set_option('fwd_split', 1)
    handle.set_option('split_fallbacks', 0)
    try:
        probs, labels = run_forward(handle, ax, co, sa, at)
        assert handle.get_option('split_fallbacks') == 0
    finally:
        handle.set_option('fwd_split', -1)
    assert np.abs(probs - p32).max() < 2e-5
    assert (labels != l32).sum() == 0 or np.abs(np.sort(p32, 1)[labels != l32, -1] - np.sort(p32, 1)[labels != l32, -2]).max() < 1e-5


def test_out_of_range_activations_fall_back_to_the_f32_kernels(split):
    """Activations are stored times 2^6 in f16 pieces: beyond |a| ~ 1000 a piece overflows.  The kernels flag it, the call repeats the
    batch on the exact-f32 kernels: same bits as the f32 path, and the counter says so."""
    g = gold('g1_random.npz')
    big = 3.0e4
    args = [g['ax'] * big, g['co'] * big, g['sa'] * big, g['atlas']]
    probs, labels = run_forward(split, *args)
    assert split.get_option('split_fallbacks') == 1
    split.set_option('fwd_split', -1)
    p32, l32 = run_forward(split, *args)
    assert np.array_equal(probs, p32) and np.array_equal(labels, l32)


def test_gather_fused_volume_path_on_the_split_kernels_equals_their_patch_path(split):
    """scs_forward_volume with fwd_split: the conv1 + conv2 + pool1 split kernel cuts the patches itself (tri-planar gather of
    base.py:293-306, zero padded): the same bits as the split patch entry on the patches the literal restatement cut."""
    from test_gpu_forward import dev
    from oracle import patches_ref as R
    g = gold('g2_volume.npz')
    vol = R.zscore_inference(g['volume']).astype(np.float32)
    probs, labels = run_forward(split, g['ax'], g['co'], g['sa'], g['atlas'])
    n = g['centers'].shape[0]
    d_pr = torch.empty((n, 15), device='cuda:0')
    d_lb = torch.empty((n,), dtype=torch.int32, device='cuda:0')
    split.forward_volume(dev(vol), vol.shape, dev(g['centers'], torch.int32), dev(g['atlas4d']), n, d_pr, d_lb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_pr.cpu().numpy(), probs) and np.array_equal(d_lb.cpu().numpy(), labels)
    assert np.abs(probs - g['probs']).max() < PROB_TOL
    assert split.get_option('split_fallbacks') == 0


def test_full_size_properties_on_the_split_path(split):
    """BASELINE configs[1] size (100 000 patches) on the split kernels: rows sum to 1, a shuffled batch gives the shuffled result bit
    for bit (persistent workgroups: a patch's result must not depend on which workgroup or iteration computed it), determinism."""
    from cnn_cort import synth
    from test_gpu_forward import dev
    n = 100000
    ax, co, sa, at = synth.patch_batch(n, seed=0)
    d = [dev(a) for a in (ax, co, sa, at)]
    st = torch.cuda.current_stream().cuda_stream
    p1 = torch.empty((n, 15), device='cuda:0')
    l1 = torch.empty((n,), dtype=torch.int32, device='cuda:0')
    split.forward(*d, n, p1, l1, st)
    perm = torch.randperm(n, device='cuda:0', generator=torch.Generator(device='cuda:0').manual_seed(1))
    p2 = torch.empty_like(p1)
    l2 = torch.empty_like(l1)
    split.forward(*[x[perm].contiguous() for x in d], n, p2, l2, st)
    p3 = torch.empty_like(p1)
    split.forward(*d, n, p3, None, st)
    torch.cuda.synchronize()
    assert torch.equal(p2, p1[perm]) and torch.equal(l2, l1[perm]) and torch.equal(p3, p1)
    assert (p1.sum(1) - 1).abs().max().item() < 1e-5
    assert split.get_option('split_fallbacks') == 0


def test_net_predict_with_fwd_split_option(weights):
    """options['fwd_split'] = 'True' -> build_model's Net runs predict / predict_proba on the split kernels, for NumPy inputs (the library's
    double-buffered host entry, scs_forward_host) and for device tensors (scs_forward): labels of the f32 net, probabilities within 2e-5."""
    import os
    import tempfile
    from cnn_cort import nets, params as P
    g = gold('g3_classes.npz')
    X = {'in1': g['ax'], 'in2': g['co'], 'in3': g['sa'], 'in4': g['atlas']}
    with tempfile.TemporaryDirectory() as d:
        opt = {'experiment': 'e', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 1, 'patch_size': (32, 32),
               'load_weights': 'False', 'device': 0}
        n32 = nets.build_model(d, dict(opt))
        n32.load_params_from(weights)
        nsp = nets.build_model(d, dict(opt, fwd_split='True'))
        nsp.load_params_from(weights)
        p32, l32 = n32.predict_proba(X), n32.predict(X)
        psp, lsp = nsp.predict_proba(X), nsp.predict(X)
        assert nsp._handle.get_option('fwd_split') == 1 and nsp._handle.get_option('split_fallbacks') == 0
        assert np.abs(psp - p32).max() < 2e-5 and np.array_equal(lsp, l32)
        assert np.abs(psp - g['probs']).max() < PROB_TOL
        Xd = {k: torch.from_numpy(np.ascontiguousarray(v)).to('cuda:0') for k, v in X.items()}
        assert np.array_equal(nsp.predict(Xd), l32)
