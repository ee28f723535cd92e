"""Option train_split on the MI355X: the training forward's convolutions on the f16 matrix cores with every f32 operand as two f16
pieces (csrc/conv_split_train.h), at the bars of the f32-MFMA step: every gradient within 1e-4 of its array's scale of the float64
oracle with the step's own ReLU / pooling decisions handed to the oracle's backward pass (tests/test_gpu_train.py), the raw conv
outputs and the batch statistics the kernels leave against the f32 kernels' on the same inputs.  The option covers the conv1 + conv2
block (conv12_split_train_kernel); conv3 / conv4 in the same form were built, passed these tests and were slower than the f32-MFMA
kernels (tools/experiments/conv_split_train34.h, profiles/NOTES_r05.md)."""
import numpy as np
import pytest
import torch

from cnn_cort import params as P
from conftest import exact_f32_form
from test_gpu_train import _step_against_gate_aware_oracle, dev, arena, fresh   # noqa: F401  (fresh: fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 5, 37, 128, 170, 256, 257, 511, 700])
def test_split_training_step_at_ragged_batch_sizes(fresh, weights, n):
    """scs_train_step with train_split = 1 against the oracle; the sizes cross the grid's switch - one workgroup per patch-tower up to
    256, 160 persistent ones beyond - and the in-kernel finalize's slices (257 .. 511: several rounds of units per workgroup;
    700: beyond fin_max_batch = 512 the statistics are finalized by the launch of their own)."""
    fresh.set_option('train_split', 1)
    worst, forced = _step_against_gate_aware_oracle(fresh, weights, n, 0)
    print('train_split, n = %d: worst relative gradient error %.3g; decisions taken from the GPU: %r' % (n, worst, forced))


@pytest.mark.parametrize('n', [5, 37])
def test_split_training_step_on_random_draws(fresh, weights, n):
    fresh.set_option('train_split', 1)
    for draw in range(1, 4):
        _step_against_gate_aware_oracle(fresh, weights, n, draw, stale_rows=False)


@pytest.mark.parametrize('n', [5, 128])
def test_split_pool1_follows_the_sign_of_gamma(fresh, weights, n):
    """As test_pool1_taken_before_batch_norm_follows_the_sign_of_gamma: negated and zero gammas in the conv2 block."""
    w = {k: [a.copy() for a in v] for k, v in weights.items()}
    for tower in ('axial', 'coronal', 'saggital'):
        name = [k for k in w if k.startswith(tower[:4]) and k.endswith('conv2_bn')]
        gamma = w[name[0]][1]
        gamma[::3] = -gamma[::3]
        gamma[1] = 0.0
    fresh.set_option('train_split', 1)
    worst, forced = _step_against_gate_aware_oracle(fresh, w, n, 0)
    print('train_split, n = %d, signed gammas: worst relative gradient error %.3g; %r' % (n, worst, forced))


@pytest.mark.parametrize('n', [3, 64, 300])
def test_split_forward_buffers_against_the_f32_kernels(weights, n):
    """The buffers the forward pass leaves - raw conv2 output, the (scale, shift) of every block's batch norm, the loss and all
    gradients - from the split form against the f32-MFMA form on the same batch: f32-grade agreement (2e-5 of each array's scale;
    the two forms round differently, nothing more), and the split form is deterministic."""
    from cnn_cort import _hip, synth
    ax, co, sa, at = synth.patch_batch(n, seed=11 + n)
    rng = np.random.RandomState(n)
    y = rng.randint(0, 15, n).astype(np.uint8)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for form in ('f32', 'split', 'split2'):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        h.set_option('train_split', 0 if form == 'f32' else 1)
        loss = h.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, seed=5, stream=st)
        z2 = [h.get_train_buffer(100 + 10 * t + i, n * c * ho * ho).copy() for t in range(3) for i, c, ho in ((2, 20, 28), (3, 40, 12), (4, 40, 10), (5, 60, 3))]
        ss = [h.get_train_buffer(200 + 10 * t + i, 2 * c).copy() for t in range(3) for i, c in ((2, 20), (3, 40), (4, 40), (5, 60))]
        out[form] = (loss, z2, ss, arena(h, 'grad'))
        h.close()
    assert np.isfinite(out['split'][0])
    assert abs(out['split'][0] - out['f32'][0]) <= 1e-5 * max(1.0, abs(out['f32'][0]))
    for k, (a, b) in enumerate(zip(out['split'][1], out['f32'][1])):             # raw outputs of conv2 .. conv5 of every tower
        assert np.abs(a - b).max() <= (2e-6 if k % 4 == 0 else 1e-5) * np.abs(b).max(), (k, np.abs(a - b).max() / np.abs(b).max())
    for a, b in zip(out['split'][2], out['f32'][2]):
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()
    # bit-for-bit repeatable
    assert out['split'][0] == out['split2'][0]
    for a, b in zip(out['split'][1], out['split2'][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(out['split'][3], out['split2'][3])


def test_split_overflow_turns_the_loss_into_nan(weights):
    """A conv1 activation beyond the f16 range of the scaled planes (|a| > 1023) cannot be represented by the split form: the step
    must not return a plausible loss (conv_split_train.h raises a flag; the workgroup that forms the batch-mean loss turns it into NaN)."""
    from cnn_cort import _hip, synth
    n = 8
    ax, co, sa, at = synth.patch_batch(n, seed=3)
    w = {k: [a.copy() for a in v] for k, v in weights.items()}
    w['axial_ch_conv1_bn'][1][:] = 5000.0          # gamma: normalised activations times 5000
    y = np.zeros(n, np.uint8)
    st = torch.cuda.current_stream().cuda_stream
    h = _hip.Handle(0)
    h.set_params(P.params_to_flat(w))
    h.set_option('train_split', 1)
    loss = h.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, seed=5, stream=st)
    h.close()
    assert not np.isfinite(loss), loss


from test_gpu_compact import toy, device_form   # noqa: E402,F401  (toy: fixture)


@pytest.mark.parametrize('augmented', [False, True])
def test_split_step_from_volumes_equals_the_step_from_patches(toy, weights, augmented):
    """With train_split the input patches are cut (and, with the opt-in augmentation, mirrored) inside the split kernel's own loads
    (conv_split_train.h: X0RowRegs): the step from volumes + voxel rows must equal the step from the materialised views bit for bit."""
    from cnn_cort import _hip
    from cnn_cort.nets import rotate_flip_views
    _, mat, cs = toy
    d = device_form(cs)
    n, lo = 40, 3
    st = torch.cuda.current_stream().cuda_stream
    views = [dev(mat[v][lo:lo + n]) for v in range(3)]
    flips = None
    if augmented:
        rng = np.random.RandomState(4)
        idx = rng.choice(n, n // 2, replace=False)
        choices = [rng.randint(0, 3, size=idx.size) for _ in range(3)]
        f = np.zeros(n, np.uint8)
        for v in range(3):
            f[idx] |= (np.array([3, 2, 1], np.uint8)[choices[v]] << (2 * v)).astype(np.uint8)
        flips = torch.from_numpy(f).to('cuda:0')
        views = [v.contiguous() for v in rotate_flip_views(views, torch.from_numpy(idx).to('cuda:0'), [torch.from_numpy(c).to('cuda:0') for c in choices])]
    out = []
    for compact in (False, True):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        h.set_option('train_split', 1)
        if compact:
            kw = {'flips': flips} if augmented else {}
            loss = h.train_step_volume(d['vols'], d['desc'], d['samples'], d['atlas'], d['y'], d['index'][lo:lo + n], n, seed=9, stream=st, **kw)
        else:
            loss = h.train_step(views[0], views[1], views[2], dev(mat[3][lo:lo + n]), dev(mat[4][lo:lo + n], torch.uint8), n, seed=9, stream=st)
        out.append((loss, arena(h, 'grad')))
        h.close()
    assert np.isfinite(out[0][0]) and out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


def test_fit_with_the_fit_split_option(weights, tmp_path):
    """options['fit_split'] = 'True' -> build_model's Net trains with libscs' option train_split (the key is NOT 'train_split': that is
    the reference's hold-out fraction): fit() on a small set walks a history within 1e-4 of the f32 net's (two epochs, same seeds) and
    learns; the handle reports the option."""
    from cnn_cort import nets, synth
    n = 512
    ax, co, sa, at = synth.patch_batch(n, seed=21)
    y = np.random.RandomState(3).randint(0, 15, n).astype(np.uint8)
    X = {'in1': ax, 'in2': co, 'in3': sa, 'in4': at}
    opt = {'experiment': 'e', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 2, 'patience': 5, 'patch_size': (32, 32),
           'load_weights': 'False', 'device': 0, 'batch_size': 64, 'seed': 4}
    hist = {}
    for key in ('f32', 'split'):
        d = tmp_path / key
        d.mkdir()
        net = nets.build_model(str(d), dict(opt, fit_split='True') if key == 'split' else dict(opt))
        net.load_params_from({k: [a.copy() for a in v] for k, v in weights.items()})
        if key == 'f32':
            exact_f32_form(net)                   # (the comparison's other side, whatever SCS_TRAIN_SPLIT says in the environment)
        net.fit(X, y)
        assert net._handle.get_option('train_split') == (1 if key == 'split' else 0)
        hist[key] = [(r['train_loss'], r['valid_loss']) for r in net.train_history_]
    assert len(hist['split']) == 2 and all(np.isfinite(v) for r in hist['split'] for v in r)
    for a, b in zip(hist['split'], hist['f32']):
        assert abs(a[0] - b[0]) <= 1e-4 * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 1e-4 * max(1.0, abs(b[1])), (hist['split'], hist['f32'])
    assert hist['split'][1][0] < hist['split'][0][0]
