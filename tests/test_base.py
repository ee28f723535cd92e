"""Host-side data path (cnn_cort.base / nifti / load_options / nets host logic) against the literal restatement
of the reference's helpers (oracle/patches_ref.py) and the committed g6 fixtures.  CPU only."""
import configparser
import json
import os
import random

import numpy as np
import pytest

from conftest import gold, GOLD
from cnn_cort import base, nifti, synth
from cnn_cort.load_options import load_options, is_true
from cnn_cort.nets import stratified_first_fold, EarlyStopping
from oracle import patches_ref as R


@pytest.mark.parametrize('mode', ['axial', 'coronal', 'saggital'])
def test_get_patches_equals_the_literal_restatement(mode):
    vol = np.random.RandomState(0).randn(20, 23, 17).astype(np.float32)
    rng = np.random.RandomState(1)
    centers = [tuple(int(v) for v in rng.randint(0, s, 3)) for s in [vol.shape] * 40]
    centers += [(0, 0, 0), (19, 22, 16), (0, 22, 0), (19, 0, 16)]          # every face / corner: zero padding
    ref = np.array(R.get_patches(vol, centers, (32, 32), mode=mode))
    got = base.get_patches(vol, centers, (32, 32), mode=mode)
    assert got.shape == (44, 32, 32) and got.dtype == vol.dtype
    assert np.array_equal(got, ref)
    c = centers[5]                                                           # centre voxel lands at [16][16]
    assert got[5, 16, 16] == vol[c]
    assert base.get_patches(vol, [], (32, 32), mode=mode).shape == (0, 32, 32)


def test_get_mask_voxels_order_and_seeded_shuffle():
    m = np.random.RandomState(2).rand(6, 7, 5) > 0.6
    ref = R.get_mask_voxels(m)
    got = base.get_mask_voxels(m)
    assert np.array_equal(got, np.asarray(ref))
    x, y, z = np.stack(got, axis=1)                                          # the reference's scatter idiom (base.py:430)
    assert m[x, y, z].all()
    ref_s = R.get_mask_voxels(m, size=10, rng=random.Random(7))
    got_s = base.get_mask_voxels(m, size=10, rng=random.Random(7))
    assert np.array_equal(got_s, np.asarray(ref_s))


def test_g6_inference_batches_match_fixture():
    g = gold('g6_toy.npz')
    vol, atlas, roi = g['volume'], g['atlas4d'], g['roi'].astype(bool)
    vn = base._zscore(vol, False)
    assert vn.dtype == np.float64                                            # integer T1 -> float64 arithmetic (base.py:358)
    centers = base.get_mask_voxels(roi)
    assert np.array_equal(centers, g['centers'])
    assert np.array_equal(base.atlas_vectors_at(atlas, centers), g['atlas_vec'])
    assert (g['atlas_vec'].sum(1) > 0).all() and (g['atlas_vec'][:, 14] == 1).any()   # zero rows became background
    for nm, mode in (('ax', 'axial'), ('co', 'coronal'), ('sa', 'saggital')):
        p = base.get_patches(vn, centers, (32, 32), mode=mode).astype(np.float32)[:, None]
        assert np.allclose(p.astype(np.float64).sum(axis=(1, 2, 3)), g['sum_' + nm], rtol=0, atol=1e-9)
        assert np.array_equal(p[:4], g['first_' + nm]) and np.array_equal(p[-4:], g['last_' + nm])


def test_g6_training_set_assembly_matches_fixture():
    g = gold('g6_toy.npz')
    vol, atlas, labels = g['volume'], g['atlas4d'], g['labels']
    im = base._zscore(vol, True)
    assert im.dtype == np.float32 or im.dtype == np.float64
    pos = base.get_mask_voxels(np.logical_and(labels > 0, labels < 15))
    neg = base.get_mask_voxels(labels == 15, size=len(pos), rng=random.Random(1234))
    cent = np.concatenate([pos, neg])
    assert np.array_equal(cent, g['train_centers'])
    xs = [np.concatenate([base.get_patches(im, pos, (32, 32), m), base.get_patches(im, neg, (32, 32), m)]) for m in ('axial', 'coronal', 'saggital')]
    y_ax = np.concatenate([base.get_patches(labels, pos, (32, 32), 'axial'), base.get_patches(labels, neg, (32, 32), 'axial')])
    atl = atlas[cent[:, 0], cent[:, 1], cent[:, 2]]
    xa, xc, xsg, xt, yt = base.generate_training_set([xs[0]], [xs[1]], [xs[2]], [atl], [y_ax], {'seed': 777, 'debug': 'False'})
    assert xa.shape[1:] == (1, 32, 32) and xa.dtype == np.float32 and yt.dtype == np.uint8
    assert np.array_equal(yt, g['train_y']) and yt.max() <= 14                   # 15 -> 0 (base.py:88-89)
    assert np.array_equal(xt, g['train_atlas'])
    assert np.allclose(xa.astype(np.float64).sum(axis=(1, 2, 3)), g['train_sum_ax'], atol=1e-9)
    assert np.allclose(xc.astype(np.float64).sum(axis=(1, 2, 3)), g['train_sum_co'], atol=1e-9)
    assert np.allclose(xsg.astype(np.float64).sum(axis=(1, 2, 3)), g['train_sum_sa'], atol=1e-9)
    assert np.array_equal(xa[:2], g['train_first_ax'])
    un = base.generate_training_set([xs[0]], [xs[1]], [xs[2]], [atl], [y_ax], {'debug': 'False'}, randomize=False)
    assert np.array_equal(un[0][:, 0], xs[0].astype(np.float32))


def test_post_processing_matches_restatement_including_its_quirk():
    g = gold('g6_toy.npz')
    out = base.post_process_mask(g['pp_in'], g['pp_mask'].astype(bool))
    assert np.array_equal(out, g['pp_out'])
    assert np.array_equal(out, R.post_process_segmentation(g['pp_in'], g['pp_mask'].astype(bool)))


def test_nifti_roundtrip_and_scaling(tmp_path):
    a = (np.random.RandomState(3).rand(5, 6, 7) * 100).astype(np.int16)
    nifti.save(a, str(tmp_path / 'a.nii.gz'), affine=np.diag([1.0, 2.0, 3.0, 1.0]))
    img = nifti.load(str(tmp_path / 'a.nii.gz'))
    assert img.get_data().dtype == np.int16 and np.array_equal(img.get_data(), a)
    assert np.allclose(img.affine, np.diag([1.0, 2.0, 3.0, 1.0]))
    f = np.random.RandomState(4).rand(4, 3, 2, 15)
    nifti.save(f, str(tmp_path / 'f.nii'), like=img)
    back = nifti.load(str(tmp_path / 'f.nii'))
    assert back.shape == (4, 3, 2, 15) and np.array_equal(back.get_data(), f)
    assert np.allclose(back.affine, img.affine)
    with pytest.raises(IOError):
        (tmp_path / 'bad.nii').write_bytes(b'\0' * 400)
        nifti.load(str(tmp_path / 'bad.nii'))


@pytest.mark.parametrize('dt', [np.float32, np.float64, np.int16, np.uint8])
@pytest.mark.parametrize('shape', [(17, 13, 11), (17, 13, 11, 5)])
def test_nifti_streaming_rows_and_sparse(tmp_path, monkeypatch, dt, shape):
    """load_rows == load()[x,y,z] and save_sparse == save(dense), across piece / member boundaries (the stream
    sizes are shrunk so that a small file spans many of them and elements straddle the pieces)."""
    import gzip
    monkeypatch.setattr(nifti, '_MEMBER_BYTES', 1 << 12)
    monkeypatch.setattr(nifti, '_GAP_ROWS', 1)                        # sparse writer: many data runs, zero runs of every length
    monkeypatch.setattr(nifti, '_MAX_ROWS_BYTES', 5 * 17 * np.dtype(dt).itemsize)      # ... and runs split over several members
    monkeypatch.setattr(nifti, '_STREAM_BYTES', 4999)
    monkeypatch.setattr(nifti, '_FEED_BYTES', 777)
    rng = np.random.default_rng(5)
    a = (rng.random(shape) * 100).astype(dt)
    vox = np.stack([rng.integers(0, shape[i], 40) for i in range(3)], 1)
    vox = np.concatenate([vox, vox[:3]])                              # duplicates: the later value wins
    vals = (rng.random((43,) + shape[3:]) * 50 + 1).astype(dt)
    dense = np.zeros(shape, dt)
    dense[vox[:, 0], vox[:, 1], vox[:, 2]] = vals                     # later duplicates win
    for ext in ('.nii.gz', '.nii'):
        fn = str(tmp_path / ('a' + ext))
        nifti.save(a, fn)
        b = nifti.load(fn).get_data()
        assert b.dtype == a.dtype and np.array_equal(a, b) and b.flags.writeable
        r = nifti.load_rows(fn, vox)
        assert r.dtype == a.dtype and np.array_equal(r, a[vox[:, 0], vox[:, 1], vox[:, 2]])
        nifti.save(dense, str(tmp_path / ('dense' + ext)))
        nifti.save_sparse(str(tmp_path / ('sparse' + ext)), shape, dt, vox, vals)
        opener = gzip.open if ext.endswith('.gz') else open
        with opener(str(tmp_path / ('dense' + ext)), 'rb') as f1, opener(str(tmp_path / ('sparse' + ext)), 'rb') as f2:
            assert f1.read() == f2.read()
    # a label map: one value almost everywhere (also a non-zero one: the post-processing quirk), through the fill-value writer
    if len(shape) == 3:
        for fillv in (0, 14):
            lab = np.full(shape, fillv, dt)
            lab[vox[:9, 0], vox[:9, 1], vox[:9, 2]] = np.arange(1, 10).astype(dt)
            nifti.save(lab, str(tmp_path / 'lab_dense.nii.gz'))
            nifti.save_mostly_constant(lab, str(tmp_path / 'lab_sparse.nii.gz'))
            with gzip.open(str(tmp_path / 'lab_dense.nii.gz'), 'rb') as f1, gzip.open(str(tmp_path / 'lab_sparse.nii.gz'), 'rb') as f2:
                assert f1.read() == f2.read()
    # a single-member stream as an external tool writes it, and a truncated one
    with gzip.open(str(tmp_path / 'a.nii.gz'), 'rb') as f:
        raw = f.read()
    with gzip.open(str(tmp_path / 'single.nii.gz'), 'wb') as f:
        f.write(raw)
    assert np.array_equal(nifti.load(str(tmp_path / 'single.nii.gz')).get_data(), a)
    assert np.array_equal(nifti.load_rows(str(tmp_path / 'single.nii.gz'), vox[:7]), a[vox[:7, 0], vox[:7, 1], vox[:7, 2]])
    blob = (tmp_path / 'single.nii.gz').read_bytes()
    (tmp_path / 'trunc.nii.gz').write_bytes(blob[:len(blob) // 2])
    with pytest.raises(IOError):
        nifti.load(str(tmp_path / 'trunc.nii.gz'))
    with pytest.raises(ValueError):
        nifti.load_rows(str(tmp_path / 'a.nii'), [[shape[0], 0, 0]])


def test_nifti_big_endian_and_scaled(tmp_path):
    """Files from other tools: big-endian voxels and scl_slope/inter (get_data() scales to float64)."""
    import struct
    a = (np.random.RandomState(9).rand(6, 5, 4) * 200).astype(np.int16)
    nifti.save(a, str(tmp_path / 'le.nii'))
    raw = bytearray((tmp_path / 'le.nii').read_bytes())
    struct.pack_into('<2f', raw, 112, 0.5, -3.0)
    (tmp_path / 'scaled.nii').write_bytes(bytes(raw))
    assert np.array_equal(nifti.load(str(tmp_path / 'scaled.nii')).get_data(), a.astype(np.float64) * 0.5 - 3.0)
    vox = np.array([[0, 0, 0], [5, 4, 3], [2, 3, 1]])
    assert np.array_equal(nifti.load_rows(str(tmp_path / 'scaled.nii'), vox), a[vox[:, 0], vox[:, 1], vox[:, 2]] * 0.5 - 3.0)
    # byte-swap the header fields this reader uses and the voxel block
    be = bytearray(raw)
    be[0:4] = struct.pack('>i', 348)
    be[40:56] = struct.pack('>8h', *struct.unpack('<8h', raw[40:56]))
    be[70:72] = struct.pack('>h', *struct.unpack('<h', raw[70:72]))
    be[108:120] = struct.pack('>3f', *struct.unpack('<3f', raw[108:120]))
    be[352:] = a.T.astype('>i2').tobytes()
    (tmp_path / 'be.nii').write_bytes(bytes(be))
    got = nifti.load(str(tmp_path / 'be.nii')).get_data()
    assert np.array_equal(got, a.astype(np.float64) * 0.5 - 3.0)


def make_subject(folder, shape=(40, 40, 40), seed=0, with_labels=False, steered=False):
    """steered: the prior file holds synth.steered_atlas (every class wins somewhere at volume level) instead of the
    probability-like toy prior (classes 0 and 2 only); returns (T1, prior as written, mask as written)."""
    os.makedirs(os.path.join(folder, 'tmp'), exist_ok=True)
    vol = synth.smooth_volume(shape, sigma=3.0, seed=seed, as_int=True)
    if steered:
        assert not with_labels
        atlas, smap = synth.steered_atlas(shape, seed=seed + 1)
        mask = (smap > 0).astype(np.float32)
    else:
        atlas = synth.toy_atlas(shape, seed=seed + 1)
        mask = (atlas[..., :13].sum(-1) > 0).astype(np.float32)
    nifti.save(vol, os.path.join(folder, 'T1.nii.gz'))
    nifti.save(atlas, os.path.join(folder, 'tmp', 'MNI_sub_probabilities.nii.gz'))
    nifti.save(mask, os.path.join(folder, 'tmp', 'MNI_subcortical_mask.nii.gz'))
    if with_labels:
        nifti.save(synth.toy_labels(shape, atlas), os.path.join(folder, 'gt_15_classes.nii.gz'))
    return vol, atlas, mask


def test_load_patch_batch_and_load_data_on_files(tmp_path):
    vol, atlas, mask = make_subject(str(tmp_path / 'train' / 's1'), with_labels=True)
    make_subject(str(tmp_path / 'train' / 's2'), seed=5, with_labels=True)
    opts = {'train_folder': str(tmp_path / 'train'), 'test_folder': str(tmp_path / 'train'), 't1_name': 'T1.nii.gz',
            'roi_name': 'gt_15_classes.nii.gz', 'patch_size': [32, 32], 'test_batch_size': 700, 'crop': 'True',
            'debug': 'False', 'seed': 3}
    t1s, names = base.load_test_names(opts)
    assert names == ['s1', 's2'] and t1s[0].endswith(os.path.join('s1', 'T1.nii.gz'))
    from scipy import ndimage
    roi = ndimage.binary_dilation(mask, iterations=10)
    ref = list(R.load_patch_batch(vol, atlas, roi, 700))
    got = list(base.load_patch_batch(t1s[0], opts))
    assert len(got) == len(ref) and len(got) > 1
    for a, b in zip(got, ref):
        for i in range(4):
            assert a[i].dtype == np.float32 and np.array_equal(a[i], b[i])
        assert np.array_equal(a[4], np.asarray(b[4]))
    x_ax, x_co, x_sa, y_ax, x_atl, nm = base.load_data(opts)
    assert len(x_ax) == 2 and x_ax[0].shape[1:] == (32, 32) and x_atl[0].shape[1] == 15
    assert x_ax[0].shape[0] == y_ax[0].shape[0] == x_atl[0].shape[0] and x_ax[0].shape[0] % 2 == 0
    out = base.generate_training_set(x_ax, x_co, x_sa, x_atl, y_ax, opts)
    assert out[0].shape[0] == x_ax[0].shape[0] + x_ax[1].shape[0] and set(np.unique(out[4])) <= set(range(15))
    os.remove(os.path.join(str(tmp_path / 'train' / 's2'), 'tmp', 'MNI_sub_probabilities.nii.gz'))
    with pytest.raises(FileNotFoundError):
        base.load_data(opts)


def test_load_options_keys_and_string_semantics(tmp_path):
    cfg = configparser.RawConfigParser()
    cfg.read_string('''
[database]
train_folder = /a
inference_folder = /b
t1_name = T1.nii.gz
roi_name = gt_15_classes.nii.gz
save_tmp = True
[model]
name = miccai2012_v1
mode = cuda0
patch_size = 32
batch_size = 256
patience = 20
net_verbose = 1
max_epochs = 100
train_split = 0.25
test_batch_size = 100000
load_weights = True
out_probabilities = False
speedup_segmentation = True
post_process = True
debug = True
''')
    o = load_options(cfg)
    ref_keys = {'experiment', 'train_folder', 'test_folder', 'output_folder', 'current_scan', 't1_name', 'roi_name',
                'out_name', 'save_tmp', 'mode', 'patch_size', 'weight_paths', 'train_split', 'max_epochs', 'patience',
                'batch_size', 'test_batch_size', 'net_verbose', 'load_weights', 'randomize_train', 'debug',
                'out_probabilities', 'post_process', 'crop'}
    assert ref_keys <= set(o)
    assert o['patch_size'] == [32, 32] and o['load_weights'] == 'True' and o['crop'] == 'True' and o['device'] == 0
    assert is_true(o['debug']) and not is_true(o['out_probabilities']) and is_true(True) and not is_true('False')


def test_train_split_is_the_first_stratified_fold():
    y = np.array([0] * 10 + [1] * 6 + [2] * 3 + [0] * 5)
    tr, va = stratified_first_fold(y, 4)
    assert len(set(tr) & set(va)) == 0 and len(tr) + len(va) == len(y)
    # per class the validation fold is the first ceil-share of that class's samples, in order of appearance
    assert list(va[y[va] == 0]) == [0, 1, 2, 3] and list(va[y[va] == 1]) == [10, 11] and list(va[y[va] == 2]) == [16]


def test_early_stopping_reproduces_the_shipped_history():
    """nets/miccai2012_v1/miccai2012_v1_history.pkl: best valid loss at epoch 42, training ended after epoch 62
    with patience 20 - replaying the history through the hook must stop exactly there."""
    hist = json.load(open(os.path.join(GOLD, 'miccai2012_v1_history.json')))
    assert len(hist) == 62 and set(hist[0]) == {'dur', 'epoch', 'train_loss', 'train_loss_best', 'valid_accuracy', 'valid_loss', 'valid_loss_best'}

    class FakeNet:
        verbose = 0
        def get_all_params_values(self): return {}
        def load_params_from(self, w): self.restored = True
    hook, net, stopped = EarlyStopping(patience=20), FakeNet(), None
    for i in range(len(hist)):
        try:
            hook(net, hist[:i + 1])
        except StopIteration:
            stopped = hist[i]['epoch']
            break
    assert hook.best_valid_epoch == 42 and stopped == 62 and net.restored


class _StubNet:
    """Deterministic stand-in for the GPU net: labels / probabilities are functions of the centre and the prior row."""

    def predict_volume(self, volume, centers, atlas4d=None, want_probs=False, batch=None, atlas_rows=None):
        c = np.asarray(centers)
        labels = (c.sum(1) + np.round(volume[c[:, 0], c[:, 1], c[:, 2]] * 3).astype(np.int64)) % 15
        probs = None
        if want_probs:
            probs = (atlas_rows + 0.01 * (labels[:, None] == np.arange(15)[None, :])).astype(np.float32)
        return labels.astype(np.int64), probs


def test_cohort_pipeline_writes_the_same_files_as_the_loop(tmp_path):
    """test_cohort (loader / GPU / writer stages overlapped) == calling test_scan per subject, file for file."""
    import gzip
    opts = {'crop': 'True', 'debug': 'False', 'test_batch_size': 5000, 'out_probabilities': 'True', 'post_process': 'True',
            'patch_size': [32, 32]}
    paths = []
    for i in range(3):
        make_subject(str(tmp_path / 'a' / ('s%d' % i)), shape=(24, 22, 20), seed=40 + i)
        make_subject(str(tmp_path / 'b' / ('s%d' % i)), shape=(24, 22, 20), seed=40 + i)
        paths.append(('a/s%d/T1.nii.gz' % i, 'b/s%d/T1.nii.gz' % i))
    net = _StubNet()
    for a, _ in paths:
        assert base.test_scan(net, str(tmp_path / a), opts) >= 0
    res = base.test_cohort(net, [str(tmp_path / b) for _, b in paths], opts)
    assert [p for p, _ in res] == [str(tmp_path / b) for _, b in paths] and all(m >= 0 for _, m in res)
    for a, b in paths:
        for f in ('out_subcortical_prob.nii.gz', 'out_subcortical_seg_prec.nii.gz'):
            with gzip.open(str(tmp_path / os.path.dirname(a) / f)) as fa, gzip.open(str(tmp_path / os.path.dirname(b) / f)) as fb:
                assert fa.read() == fb.read()
    # a failing subject surfaces as an exception, not as a silently missing file
    os.remove(str(tmp_path / 'b' / 's1' / 'tmp' / 'MNI_sub_probabilities.nii.gz'))
    with pytest.raises(FileNotFoundError):
        base.test_cohort(net, [str(tmp_path / b) for _, b in paths], opts)
    assert base.test_cohort(net, [], opts) == []


def test_rotate_flip_augmentation_equals_the_reference_formulas():
    """nets.rotate_flip_views == the three variants Rotate_batch_Iterator.transform builds (nets.py:64-84), with the
    random draws injected."""
    import torch
    from cnn_cort.nets import rotate_flip_views
    rng = np.random.RandomState(8)
    views = [rng.rand(10, 1, 32, 32).astype(np.float32) for _ in range(3)]
    indices = rng.choice(10, 5, replace=False)
    choices = [rng.randint(0, 3, size=5) for _ in range(3)]
    got = rotate_flip_views([torch.from_numpy(v) for v in views], torch.from_numpy(indices),
                            [torch.from_numpy(c) for c in choices])
    for v in range(3):
        x = views[v][indices]
        rotate_90 = x[:, :, ::-1, :].transpose(0, 1, 3, 2)
        rotate_180 = rotate_90[:, :, ::-1, :].transpose(0, 1, 3, 2)
        augmented = np.stack([rotate_180, x[:, :, :, ::-1], rotate_180[:, :, :, ::-1]], axis=1)
        want = views[v].copy()
        want[indices] = np.stack([augmented[i, choices[v][i]] for i in range(5)])
        assert np.array_equal(got[v].numpy(), want)
        assert not np.array_equal(want, views[v])


@pytest.mark.parametrize('seed', range(6))
def test_post_processing_fast_path_equals_the_literal_procedure(seed):
    """Bounding-box labelling vs the literal whole-volume procedure on random label maps, including classes that are
    absent, classes without any atlas overlap (quirk: they paint their complement) and single-class volumes."""
    rng = np.random.RandomState(100 + seed)
    shape = (18, 16, 14)
    if seed == 5:
        seg = np.full(shape, 3, np.int16)                      # one class fills the volume
    else:
        seg = np.zeros(shape, np.int16)
        for _ in range(30):
            c = rng.randint(1, 15 if seed % 2 else 11)          # even seeds leave classes 11..14 empty
            x, y, z = [rng.randint(0, s - 3) for s in shape]
            seg[x:x + rng.randint(1, 4), y:y + rng.randint(1, 4), z:z + rng.randint(1, 4)] = c
    atlas = rng.rand(*shape) < (0.0 if seed == 4 else 0.3)       # seed 4: no overlap at all
    want = R.post_process_segmentation(seg, atlas)
    got = base.post_process_mask(seg, atlas)
    assert got.dtype == want.dtype and np.array_equal(got, want)
    nz = np.nonzero(seg)                                         # with the box test_scan passes: where the labels can be non-zero
    box = tuple(slice(int(i.min()), int(i.max()) + 1) for i in nz)
    assert np.array_equal(base.post_process_mask(seg, atlas, box), want)


def test_compact_training_set_is_the_materialised_one(tmp_path):
    """load_data_compact / generate_training_set_compact under the same seed describe exactly the arrays that load_data /
    generate_training_set build: same voxels in the same order, same prior rows, same labels, same permutation."""
    make_subject(str(tmp_path / 'train' / 's1'), shape=(40, 44, 36), with_labels=True)
    make_subject(str(tmp_path / 'train' / 's2'), shape=(36, 40, 44), seed=5, with_labels=True)
    opts = {'train_folder': str(tmp_path / 'train'), 't1_name': 'T1.nii.gz', 'roi_name': 'gt_15_classes.nii.gz',
            'patch_size': [32, 32], 'debug': 'False', 'seed': 11}
    x_ax, x_co, x_sa, y_ax, x_atl, _ = base.load_data(opts)
    x_ax, x_co, x_sa, y_ax = [[np.asarray(a) for a in l] for l in (x_ax, x_co, x_sa, y_ax)]     # real arrays: the materialising code
    ref = base.generate_training_set(x_ax, x_co, x_sa, x_atl, y_ax, opts)
    assert all(isinstance(a, np.ndarray) for a in ref)
    data = base.load_data_compact(opts)
    cs = base.generate_training_set_compact(data, opts)
    assert len(cs) == ref[0].shape[0] and cs.samples.shape == (len(cs), 4) and len(cs.volumes) == 2
    got = cs.materialize()
    for a, b in zip(got, ref):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    part = cs.materialize(7, 50)
    assert all(np.array_equal(p, r[7:50]) for p, r in zip(part, ref))
    offs, dims = cs.descriptors()
    assert offs == [0, 40 * 44 * 36] and dims == [(40, 44, 36), (36, 40, 44)]
    # not randomised: subject order, positives before negatives
    cs2 = base.generate_training_set_compact(data, opts, randomize=False)
    ref2 = base.generate_training_set(x_ax, x_co, x_sa, x_atl, y_ax, opts, randomize=False)
    assert all(np.array_equal(a, b) for a, b in zip(cs2.materialize(), ref2))


def test_compact_training_set_refuses_inconsistent_geometry():
    """The device reads label volumes with the image volumes' geometry and follows sample / index rows: what would be an
    out-of-bounds read there is a ValueError at construction."""
    from cnn_cort.base import CompactTrainingSet
    vol = [np.zeros((8, 9, 10), np.float32)]
    samples = np.array([[0, 1, 2, 3], [0, 4, 5, 6]], np.int32)
    atlas = np.full((2, 15), 1 / 15.0, np.float32)
    y = np.zeros(2, np.uint8)
    ok = CompactTrainingSet(vol, [np.zeros((8, 9, 10), np.uint8)], samples, atlas, y, np.array([1, 0], np.int32))
    assert len(ok) == 2
    with pytest.raises(ValueError):
        CompactTrainingSet(vol, [np.zeros((1, 1, 1), np.uint8)], samples, atlas, y, np.array([1, 0], np.int32))
    with pytest.raises(ValueError):
        CompactTrainingSet(vol, [np.zeros((8, 9, 10), np.uint8)], np.array([[1, 1, 2, 3], [0, 4, 5, 6]], np.int32), atlas, y, np.array([0], np.int32))
    with pytest.raises(ValueError):
        CompactTrainingSet(vol, [np.zeros((8, 9, 10), np.uint8)], samples, atlas, y, np.array([2], np.int32))


def test_unchanged_call_sequence_stays_lazy_and_materialises_to_the_reference_arrays(tmp_path):
    """train_model.py:36-60 unchanged: load_data returns LazyPatches (the reference's per-subject [n, 32, 32] arrays in shape, dtype,
    len and indexing - equal to the literal get_patches lists the moment they are turned into arrays), generate_training_set
    recognises them and returns lazy [N, 1, 32, 32] views of ONE CompactTrainingSet + real prior / label arrays, equal to what the
    materialising code returns for real arrays under the same seed; real arrays in -> real arrays out."""
    vol, atlas, mask = make_subject(str(tmp_path / 'train' / 's1'), shape=(40, 44, 36), with_labels=True)
    make_subject(str(tmp_path / 'train' / 's2'), shape=(36, 40, 44), seed=5, with_labels=True)
    opts = {'train_folder': str(tmp_path / 'train'), 't1_name': 'T1.nii.gz', 'roi_name': 'gt_15_classes.nii.gz',
            'patch_size': [32, 32], 'debug': 'False', 'seed': 11}
    x_ax, x_co, x_sa, y_ax, x_atl, names = base.load_data(opts)
    assert all(isinstance(a, base.LazyPatches) for l in (x_ax, x_co, x_sa, y_ax) for a in l)
    a0 = x_ax[0]
    n0 = len(a0)
    assert a0.shape == (n0, 32, 32) and a0.dtype == np.float32 and a0.ndim == 3 and y_ax[0].shape == (n0, 32, 32)
    # the literal per-voxel procedure of the reference on the same voxels (oracle/patches_ref.py)
    vn = R.zscore_training(vol)
    lit = np.array(R.get_patches(vn, [tuple(c) for c in a0.centers], (32, 32), mode='axial')).astype(np.float32)   # (float64 under NumPy 2 promotion)
    assert np.array_equal(np.asarray(a0), lit) and np.array_equal(a0[5], lit[5]) and np.array_equal(a0[3:9], lit[3:9])
    assert np.array_equal(a0[np.array([7, 2])], lit[[7, 2]]) and np.array_equal(a0[:, 16, 16], lit[:, 16, 16])
    assert np.array_equal(np.concatenate(x_co, axis=0).astype('float32')[:n0], np.asarray(x_co[0]))   # what generate_training_set does
    lazy = base.generate_training_set(x_ax, x_co, x_sa, x_atl, y_ax, opts)
    assert all(isinstance(v, base.LazyTrainingView) for v in lazy[:3]) and lazy[0].compact is lazy[1].compact is lazy[2].compact
    assert isinstance(lazy[3], np.ndarray) and lazy[3].dtype == np.float32 and isinstance(lazy[4], np.ndarray) and lazy[4].dtype == np.uint8
    real_in = [[np.asarray(a) for a in l] for l in (x_ax, x_co, x_sa)]
    ref = base.generate_training_set(real_in[0], real_in[1], real_in[2], x_atl, [np.asarray(a) for a in y_ax], opts)
    assert all(isinstance(a, np.ndarray) for a in ref)
    N = ref[0].shape[0]
    for got, want in zip(lazy, ref):
        assert tuple(got.shape) == want.shape and got.dtype == want.dtype and np.array_equal(np.asarray(got), want)
    assert len(lazy[0]) == N and np.array_equal(lazy[1][10:20], ref[1][10:20]) and np.array_equal(lazy[2][4], ref[2][4])
    # not randomised, and a mixed input (one real array) falls back to the materialising code
    un = base.generate_training_set(x_ax, x_co, x_sa, x_atl, y_ax, opts, randomize=False)
    assert np.array_equal(np.asarray(un[0])[:n0, 0], lit)
    mixed = base.generate_training_set([np.asarray(x_ax[0]), x_ax[1]], x_co, x_sa, x_atl, y_ax, opts)
    assert isinstance(mixed[0], np.ndarray) and np.array_equal(mixed[0], ref[0])
    # the set Net.fit trains on carries the prior rows and labels of the CALL (edited labels are honoured)
    cs = base.compact_from_views({'in1': lazy[0], 'in2': lazy[1], 'in3': lazy[2], 'in4': lazy[3]}, lazy[4])
    assert cs is not None and np.array_equal(cs.y[cs.index], lazy[4]) and np.array_equal(cs.atlas[cs.index], lazy[3])
    y_edit = lazy[4].copy()
    y_edit[y_edit == 3] = 4
    cs2 = base.compact_from_views({'in1': lazy[0], 'in2': lazy[1], 'in3': lazy[2], 'in4': lazy[3]}, y_edit)
    assert np.array_equal(cs2.y[cs2.index], y_edit)
    assert base.compact_from_views({'in1': ref[0], 'in2': lazy[1], 'in3': lazy[2], 'in4': lazy[3]}, lazy[4]) is None


def test_steered_atlas_is_a_deterministic_fifteen_class_prior():
    """synth.steered_atlas (the prior of the 15-class synthetic subject of the GPU tests and of bench.py's `dice` block): 14 blobs
    whose rows are the steering vectors of cnn_cort/steer15.json (tools/make_steer.py), the background vector between them, all-zero
    rows far away (the zero-row -> background rule), a structure map that names the blobs - and the same bytes on every call."""
    import json
    from conftest import PKG
    d = json.load(open(os.path.join(PKG, 'cnn_cort', 'steer15.json')))
    v = np.asarray(d['vectors'], np.float32)
    assert v.shape == (15, 15) and np.isfinite(v).all() and min(d['held_out_hit_rate']) > 0.9
    M = synth.steer_matrix()
    assert np.array_equal(M[14], v[0]) and np.array_equal(M[:14], v[1:])          # prior channel k steers to class k + 1, channel 14 to background
    shape = (36, 34, 30)
    p1, m1 = synth.steered_atlas(shape, seed=5)
    p2, m2 = synth.steered_atlas(shape, seed=5)
    assert p1.dtype == np.float32 and p1.shape == shape + (15,) and m1.dtype == np.uint8
    assert np.array_equal(p1, p2) and np.array_equal(m1, m2)
    assert set(np.unique(m1)) <= set(range(15)) and len(np.unique(m1)) >= 10        # most blobs own voxels
    zero = np.abs(p1).sum(-1) == 0
    assert zero.any() and (m1[zero] == 0).all()                                     # rows far from every blob stay all zero
    core = m1 > 0
    k = m1[core].astype(int) - 1
    # deep inside a blob the row is that blob's vector (the blend weight saturates at 1 well before the centre)
    rows = p1[core]
    exact = np.all(rows == M[k], axis=1)
    assert exact.any()
    between = (~zero) & (~core)
    assert between.any() and np.allclose(p1[between][np.abs(p1[between] - M[14]).sum(-1) < 1e-6], M[14])
