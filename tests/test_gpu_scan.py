"""End-to-end drop-in test on the MI355X: build_model() + test_scan() on a synthetic subject folder, compared with
the reference's own procedure restated on the CPU (literal load_patch_batch -> float64 oracle forward -> scatter
-> post-process).  Labels must be identical voxel for voxel, probabilities within 1e-4."""
import os

import numpy as np
import pytest

from conftest import PKG
from cnn_cort import base, nifti
from cnn_cort.nets import build_model
from oracle import np_oracle as O, patches_ref as R
from test_base import make_subject

pytestmark = pytest.mark.gpu


def options(folder, **kw):
    o = {'experiment': 'miccai2012_v1', 'train_folder': folder, 'test_folder': folder, 't1_name': 'T1.nii.gz',
         'roi_name': 'gt_15_classes.nii.gz', 'patch_size': [32, 32], 'test_batch_size': 3000, 'crop': 'True',
         'debug': 'True', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 2, 'patience': 20, 'batch_size': 128,
         'load_weights': 'True', 'out_probabilities': 'True', 'post_process': 'True', 'mode': 'cuda0', 'device': 0}
    o.update(kw)
    return o


def test_test_scan_matches_the_cpu_procedure(tmp_path, weights):
    from scipy import ndimage
    folder = str(tmp_path / 'test')
    vol, atlas, mask = make_subject(os.path.join(folder, 'subj1'), shape=(36, 36, 36), seed=11)
    opts = options(folder)
    net = build_model(os.path.join(PKG, 'nets'), opts)           # loads nets/miccai2012_v1/miccai2012_v1.{f32,json}
    t1s, names = base.load_test_names(opts)
    minutes = base.test_scan(net, t1s[0], opts)
    assert minutes >= 0
    sub = os.path.join(folder, 'subj1')
    seg = nifti.load(os.path.join(sub, 'out_subcortical_seg_prec.nii.gz')).get_data()
    prob = nifti.load(os.path.join(sub, 'out_subcortical_prob.nii.gz')).get_data()
    assert seg.dtype == vol.dtype and seg.shape == vol.shape and prob.shape == vol.shape + (15,)

    # raw segmentation (no post-processing) from a second call
    opts2 = options(folder, post_process='False', out_probabilities='False')
    base.test_scan(net, t1s[0], opts2)
    rawseg = nifti.load(os.path.join(sub, 'out_subcortical_rawseg.nii.gz')).get_data()

    # the reference's procedure on the CPU (literal load_patch_batch -> oracle), on a fixed subset of the ROI
    # voxels (the float64 oracle needs ~5 ms per patch; the ROI has tens of thousands)
    roi = ndimage.binary_dilation(mask, iterations=10)
    centers_all = R.get_mask_voxels(roi)
    assert (rawseg[~roi] == 0).all() and (prob[~roi] == 0).all()            # nothing written outside the ROI
    pick = np.random.RandomState(0).choice(len(centers_all), 1500, replace=False)
    centers = [centers_all[i] for i in sorted(pick)]
    vn = R.zscore_inference(vol)
    views = [np.stack([np.array(R.get_patches(vn, centers, (32, 32), mode=m)).astype(np.float32)], axis=1)
             for m in ('axial', 'coronal', 'saggital')]
    p = O.forward(weights, views[0], views[1], views[2], R.atlas_vectors(atlas, centers))
    x, y, z = np.stack(centers, axis=1)
    srt = np.sort(p, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert np.abs(prob[x, y, z, :] - p).max() < 1e-4
    assert np.array_equal(rawseg[x, y, z][clear], p.argmax(1)[clear])
    # per-class Dice between the GPU labels and the CPU oracle labels on those voxels: exactly 1
    for c in range(15):
        a, b = rawseg[x, y, z][clear] == c, p.argmax(1)[clear] == c
        if a.sum() + b.sum():
            assert 2.0 * (a & b).sum() / (a.sum() + b.sum()) == 1.0
    # post-processing of the GPU labels equals the literal restatement (quirk included)
    assert np.array_equal(seg, R.post_process_segmentation(rawseg, mask.astype(bool)))


def test_predict_api_accepts_numpy_and_device_tensors(weights):
    import torch
    from conftest import gold
    g = gold('g3_classes.npz')
    net = build_model(os.path.join(PKG, 'nets'), options('/nonexistent'))
    X = {'in1': g['ax'], 'in2': g['co'], 'in3': g['sa'], 'in4': g['atlas']}
    p = net.predict_proba(X)
    y = net.predict(X)
    assert p.dtype == np.float32 and p.shape == (120, 15) and y.dtype == np.int64
    assert np.abs(p - g['probs']).max() < 1e-4 and np.array_equal(y, g['probs'].argmax(1))
    Xd = {k: torch.from_numpy(v).cuda() for k, v in X.items()}
    assert np.array_equal(net.predict_proba(Xd), p)
    with pytest.raises(ValueError):
        net.predict({'in1': g['ax'][:, :, :16], 'in2': g['co'], 'in3': g['sa'], 'in4': g['atlas']})
    with pytest.raises(IOError):
        build_model(str(os.path.join(PKG, 'csrc')), options('/nonexistent'))      # no weights there: loud failure


def test_volume_rows_equals_volume_4d(weights):
    """scs_forward_volume_rows (host-gathered prior rows) and scs_forward_volume (device gather from the 4-D prior)
    are the same computation: identical labels and bit-identical probabilities, incl. zero-prior rows."""
    from cnn_cort import synth
    shape = (40, 36, 32)
    vol = base._zscore(synth.smooth_volume(shape, sigma=3.0, seed=21, as_int=True), False).astype(np.float32)
    atlas = synth.toy_atlas(shape, seed=22).astype(np.float32)
    atlas[:6] = 0                                                   # a slab without any prior -> background rule
    rng = np.random.RandomState(23)
    centers = np.stack([rng.randint(0, s, 700) for s in shape], axis=1)
    net = build_model(os.path.join(PKG, 'nets'), options('/nonexistent'))
    la, pa = net.predict_volume(vol, centers, atlas4d=atlas, want_probs=True, batch=256)
    lb, pb = net.predict_volume(vol, centers, atlas_rows=base.atlas_vectors_at(atlas, centers), want_probs=True, batch=300)
    assert np.array_equal(la, lb) and np.array_equal(pa, pb)
    with pytest.raises(ValueError):
        net.predict_volume(vol, centers, want_probs=True)
    with pytest.raises(ValueError):
        net.predict_volume(vol, centers, atlas_rows=np.zeros((3, 15), np.float32))


@pytest.mark.parametrize('shape,iters', [((23, 17, 31), 1), ((40, 44, 36), 10), ((9, 64, 5), 3), ((33, 20, 18), 0)])
def test_gpu_dilation_equals_scipy(shape, iters):
    """Net.dilate_mask == scipy.ndimage.binary_dilation(mask, iterations) bit for bit (6-neighbour cross, border 0), on
    masks that touch the volume faces and on non-cubic, non-multiple-of-4 shapes (base.py:369)."""
    from scipy import ndimage
    from cnn_cort.nets import Net
    rng = np.random.RandomState(sum(shape) + iters)
    mask = rng.rand(*shape) < 0.01
    mask[0, 0, 0] = mask[-1, -1, -1] = True
    mask[shape[0] // 2, 0, shape[2] - 1] = True
    net = Net(device=0)
    got = net.dilate_mask(mask.astype(np.float32) * 3.0, iters)          # any non-zero value counts as set (the mask file is float)
    want = ndimage.binary_dilation(mask, iterations=iters) if iters else mask
    assert got.dtype == bool and got.shape == mask.shape and np.array_equal(got, want)
