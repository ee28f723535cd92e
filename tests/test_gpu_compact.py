"""Gather-fused training path on the MI355X: a toy subject folder goes through load_data_compact /
generate_training_set_compact and is trained from volumes + voxel rows; results must be bit-identical to the
host-materialised path (load_data / generate_training_set -> [N,1,32,32] arrays) and agree with the float64 oracle on
batches cut by the literal restatement of the reference's get_patches (oracle/patches_ref.py)."""
import os

import numpy as np
import pytest
import torch

from cnn_cort import _hip, base, dist, params as P
from oracle import np_oracle as O, patches_ref as R
from test_base import make_subject
from test_gpu_train import GRAD_RTOL, arena, dev, pack_masks, _random_masks

pytestmark = pytest.mark.gpu


_LAZY = {}


@pytest.fixture(scope='module')
def toy(tmp_path_factory):
    root = tmp_path_factory.mktemp('compact')
    make_subject(str(root / 'train' / 's1'), shape=(40, 44, 36), with_labels=True)
    make_subject(str(root / 'train' / 's2'), shape=(36, 40, 44), seed=5, with_labels=True)
    opts = {'train_folder': str(root / 'train'), 't1_name': 'T1.nii.gz', 'roi_name': 'gt_15_classes.nii.gz',
            'patch_size': [32, 32], 'debug': 'False', 'seed': 11}
    x_ax, x_co, x_sa, y_ax, x_atl, _ = base.load_data(opts)
    lazy = base.generate_training_set(x_ax, x_co, x_sa, x_atl, y_ax, opts)          # the unchanged call sequence: lazy views
    mat = base.generate_training_set(*[[np.asarray(a) for a in l] for l in (x_ax, x_co, x_sa)], x_atl, [np.asarray(a) for a in y_ax], opts)
    cs = base.generate_training_set_compact(base.load_data_compact(opts), opts)
    _LAZY['views'] = lazy
    return opts, mat, cs


def device_form(cs):
    offs, dims = cs.descriptors()
    desc = np.zeros((len(offs), 3), np.int64)
    d32 = desc.view(np.int32).reshape(len(offs), 6)
    for i, (o, d) in enumerate(zip(offs, dims)):
        desc[i, 0] = o
        d32[i, 2:5] = d
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0')
    return dict(vols=t(np.concatenate([v.ravel() for v in cs.volumes])), desc=t(desc), samples=t(cs.samples), atlas=t(cs.atlas),
                y=t(cs.y), labvols=t(np.concatenate([v.ravel() for v in cs.labels])), index=t(cs.index))


def test_labels_are_read_on_the_device(toy, handle):
    _, _, cs = toy
    d = device_form(cs)
    y = torch.full((cs.samples.shape[0],), 99, dtype=torch.uint8, device='cuda:0')
    handle.gather_labels(d['labvols'], d['desc'], d['samples'], cs.samples.shape[0], y, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), cs.y) and set(np.unique(cs.y)) <= set(range(15)) and (cs.y == 0).any()


@pytest.mark.parametrize('n', [48, 37])
def test_step_from_volumes_equals_step_from_patches_and_the_oracle(toy, weights, n):
    _, mat, cs = toy
    d = device_form(cs)
    lo = 5
    rows = cs.index[lo:lo + n]
    ax, co, sa, at, y = [a[lo:lo + n] for a in mat]
    masks = _random_masks(np.random.RandomState(n), n)
    dm = dev(pack_masks(masks), torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for compact in (False, True):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        if compact:
            loss = h.train_step_volume(d['vols'], d['desc'], d['samples'], d['atlas'], d['y'], d['index'][lo:lo + n], n, masks=dm, stream=st)
        else:
            loss = h.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, masks=dm, stream=st)
        out.append((loss, arena(h, 'grad'), arena(h, 'param')))
        h.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])       # bit-identical
    # the literal procedure of the reference: per-voxel slices of the padded volume (oracle/patches_ref.get_patches)
    lit = []
    for mode in ('axial', 'coronal', 'saggital'):
        x = np.zeros((n, 32, 32), np.float32)
        for i, (v, cx, cy, cz) in enumerate(cs.samples[rows]):
            x[i] = np.asarray(R.get_patches(cs.volumes[v], [(int(cx), int(cy), int(cz))], (32, 32), mode=mode)[0], np.float32)
        lit.append(x[:, None])
    assert all(np.array_equal(a, b) for a, b in zip(lit, (ax, co, sa)))
    loss_ref, grads_ref, _, _ = O.train_forward_backward(weights, lit[0], lit[1], lit[2], cs.atlas[rows], cs.y[rows], masks)
    assert np.isfinite(loss_ref), loss_ref
    assert abs(out[1][0] - loss_ref) <= 2e-4 * max(1.0, abs(loss_ref))
    got = P.flat_to_params(out[1][1])
    for name, shapes in P.LAYER_SPEC:
        for i, _ in enumerate(shapes):
            ref = np.asarray(grads_ref[name][i])
            scale = max(np.abs(ref).max(), 1e-6)
            assert np.abs(got[name][i] - ref).max() / scale <= 2 * GRAD_RTOL, name      # smooth toy volumes: small BN variances amplify rounding


def test_eval_from_volumes_equals_eval_from_patches(toy, handle):
    _, mat, cs = toy
    d = device_form(cs)
    n = len(cs)
    st = torch.cuda.current_stream().cuda_stream
    a = handle.eval_batch(dev(mat[0]), dev(mat[1]), dev(mat[2]), dev(mat[3]), dev(mat[4], torch.uint8), n, stream=st)
    b = handle.eval_batch_volume(d['vols'], d['desc'], d['samples'], d['atlas'], d['y'], d['index'], n, stream=st)
    assert a == b
    handle.set_chunk(64)                                     # several chunks, with and without an index
    try:
        c = handle.eval_batch_volume(d['vols'], d['desc'], d['samples'], d['atlas'], d['y'], d['index'], n, stream=st)
        e = handle.eval_batch_volume(d['vols'], d['desc'], d['samples'], d['atlas'], d['y'], None, n, stream=st)
        inv = np.argsort(cs.index)
        f = handle.eval_batch(dev(mat[0][inv]), dev(mat[1][inv]), dev(mat[2][inv]), dev(mat[3][inv]), dev(mat[4][inv], torch.uint8), n, stream=st)
    finally:
        handle.set_chunk(32768)
    assert c[1] == a[1] and abs(c[0] - a[0]) <= 1e-9 * abs(a[0])
    assert e[1] == f[1] and abs(e[0] - f[0]) <= 1e-9 * abs(f[0])


def test_fit_compact_streamed_and_resident_agree(toy, weights):
    """fit() on the compact set, on the materialised set uploaded once, and on the materialised set streamed in slices of two
    mini-batches walk through identical batches with identical dropout seeds: the histories are bit-identical; the last batch
    of every epoch is ragged and train_loss is the batch-size-weighted mean (nolearn)."""
    from cnn_cort.nets import Net, stratified_first_fold
    _, mat, cs = toy
    X = {'in1': mat[0], 'in2': mat[1], 'in3': mat[2], 'in4': mat[3]}
    # the UNCHANGED call sequence of train_model.py:36-60: the dict of lazy views generate_training_set returned for load_data's
    # lists - fit() must take the gather-fused path (no patch is cut on the host) and walk the same batches
    lazy = _LAZY['views']
    XL = {'in1': lazy[0], 'in2': lazy[1], 'in3': lazy[2], 'in4': lazy[3]}
    hist = {}
    for mode in ('compact', 'resident', 'streamed', 'unchanged sequence'):
        net = Net(max_epochs=2, patience=20, eval_size=0.25, batch_size=48, verbose=0, device=0, seed=4,
                  fit_slice=96, fit_resident_bytes=(2 << 30) if mode == 'resident' else 0)
        net.load_params_from({k: [a.copy() for a in v] for k, v in weights.items()})
        if mode == 'unchanged sequence':
            real_get_patches = base.get_patches
            base.get_patches = lambda *a, **k: (_ for _ in ()).throw(AssertionError('fit() cut patches on the host'))
            try:
                net.fit(XL, lazy[4])
            finally:
                base.get_patches = real_get_patches
        else:
            net.fit(cs if mode == 'compact' else X, None if mode == 'compact' else mat[4])
        hist[mode] = [(r['train_loss'], r['valid_loss'], r['valid_accuracy']) for r in net.train_history_]
        final = net.get_all_params_values()
    assert hist['compact'] == hist['resident'] == hist['streamed'] == hist['unchanged sequence'], hist
    # weighted mean over ragged batches: replay epoch 1 batch by batch with the same dropout seeds on a fresh handle
    tr, va = stratified_first_fold(mat[4], 4)
    assert len(tr) % 48 != 0
    h = _hip.Handle(0)
    h.set_params(P.params_to_flat(weights))
    st = torch.cuda.current_stream().cuda_stream
    tot, cnt = 0.0, 0
    for step, lo in enumerate(range(0, len(tr), 48)):
        idx = tr[lo:lo + 48]
        loss = h.train_step(dev(mat[0][idx]), dev(mat[1][idx]), dev(mat[2][idx]), dev(mat[3][idx]), dev(mat[4][idx], torch.uint8), len(idx),
                            seed=(4 << 32) + (step + 1) * 131, stream=st)
        h.adam_apply(stream=st)
        tot += loss * len(idx)
        cnt += len(idx)
    h.close()
    assert abs(hist['resident'][0][0] - tot / cnt) <= 1e-6 * abs(tot / cnt)


def test_out_of_range_rows_and_volume_numbers_are_clamped(toy, weights):
    """Nothing a caller passes can make the gather kernels leave their buffers: index entries beyond the sample table and
    volume numbers beyond the volume table are clamped on the device (voxels outside a volume already read as zero padding)."""
    _, _, cs = toy
    d = device_form(cs)
    n, N = 16, cs.samples.shape[0]
    bad_index = torch.tensor([N + 5, -3, 2 ** 30, 0] * 4, dtype=torch.int32, device='cuda:0')
    clamped = torch.tensor([N - 1, 0, N - 1, 0] * 4, dtype=torch.int32, device='cuda:0')
    smp = cs.samples.copy()
    smp[:, 0] = np.where(np.arange(N) % 2 == 0, smp[:, 0], smp[:, 0] + 2)          # volume 2, 3: beyond the table of 2
    smp[5, 1:] = [10 ** 6, -7, 3]                                                   # a voxel far outside its volume
    ref = cs.samples.copy()
    ref[:, 0] = np.minimum(smp[:, 0], 1)
    ref[5, 1:] = smp[5, 1:]
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for samples, index in ((smp, bad_index), (ref, clamped)):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        t = torch.from_numpy(np.ascontiguousarray(samples)).to('cuda:0')
        loss = h.train_step_volume(d['vols'], d['desc'], t, d['atlas'], d['y'], index, n, seed=3, stream=st)
        ev = h.eval_batch_volume(d['vols'], d['desc'], t, d['atlas'], d['y'], index, n, stream=st)
        out.append((loss, arena(h, 'grad'), ev))
        h.close()
    assert np.isfinite(out[0][0]) and np.isfinite(out[0][1]).all()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


def test_augmented_step_and_fit_from_volumes_equal_the_materialised_ones(toy, weights):
    """The opt-in rotate / flip augmentation: as mirror flags inside the gather (compact path) it produces the same step as
    rotate_flip_views applied to the materialised batch, and fit(augment=True) walks the same history on both forms."""
    from cnn_cort.nets import Net, rotate_flip_views
    _, mat, cs = toy
    d = device_form(cs)
    n, lo = 40, 3
    rng = np.random.RandomState(4)
    idx = rng.choice(n, n // 2, replace=False)
    choices = [rng.randint(0, 3, size=idx.size) for _ in range(3)]
    flips = np.zeros(n, np.uint8)
    for v in range(3):
        flips[idx] |= (np.array([3, 2, 1], np.uint8)[choices[v]] << (2 * v)).astype(np.uint8)
    views = rotate_flip_views([dev(mat[v][lo:lo + n]) for v in range(3)], torch.from_numpy(idx).to('cuda:0'),
                              [torch.from_numpy(c).to('cuda:0') for c in choices])
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for compact in (False, True):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        if compact:
            loss = h.train_step_volume(d['vols'], d['desc'], d['samples'], d['atlas'], d['y'], d['index'][lo:lo + n], n, seed=9, stream=st,
                                       flips=torch.from_numpy(flips).to('cuda:0'))
        else:
            loss = h.train_step(views[0].contiguous(), views[1].contiguous(), views[2].contiguous(), dev(mat[3][lo:lo + n]),
                                dev(mat[4][lo:lo + n], torch.uint8), n, seed=9, stream=st)
        out.append((loss, arena(h, 'grad')))
        h.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    assert (flips != 0).sum() == n // 2
    hist = []
    for X, y in ((cs, None), ({'in1': mat[0], 'in2': mat[1], 'in3': mat[2], 'in4': mat[3]}, mat[4])):
        net = Net(max_epochs=2, patience=20, eval_size=0.25, batch_size=48, verbose=0, device=0, seed=6, augment=True)
        net.load_params_from({k: [a.copy() for a in v] for k, v in weights.items()})
        net.fit(X, y)
        hist.append([(r['train_loss'], r['valid_loss']) for r in net.train_history_])
    assert hist[0] == hist[1]
    plain = Net(max_epochs=1, patience=20, eval_size=0.25, batch_size=48, verbose=0, device=0, seed=6)
    plain.load_params_from({k: [a.copy() for a in v] for k, v in weights.items()})
    plain.fit(cs)
    assert plain.train_history_[0]['train_loss'] != hist[0][0][0]                  # the augmentation does change the batches
