"""End-to-end drop-in test on the MI355X: build_model() + test_scan() on a synthetic subject folder, compared with
the reference's own procedure restated on the CPU (literal load_patch_batch -> float64 oracle forward -> scatter
-> post-process).  Labels must be identical voxel for voxel, probabilities within 1e-4."""
import os

import numpy as np
import pytest

from conftest import PKG, exact_f32_form
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
    """36^3 subject whose prior is synth.steered_atlas: all 15 classes appear at volume level, so the label scatter, the per-class
    Dice and post_process_segmentation (base.py:425-431, 460-480) run with 14 live structures, not the 2 classes smooth volumes
    with probability-like priors give."""
    from scipy import ndimage
    from cnn_cort import synth
    folder = str(tmp_path / 'test')
    vol, atlas, mask = make_subject(os.path.join(folder, 'subj1'), shape=(36, 36, 36), seed=11, steered=True)
    smap = synth.steered_atlas((36, 36, 36), seed=12)[1]
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
    live = np.unique(rawseg)
    assert len(live) >= 13, 'classes at volume level: %s' % live

    # the reference's procedure on the CPU (literal load_patch_batch -> oracle), on a fixed subset of the ROI
    # voxels (the float64 oracle needs ~5 ms per patch; the ROI has tens of thousands): up to 70 voxels of every structure
    # blob and 700 drawn from the whole ROI
    roi = ndimage.binary_dilation(mask, iterations=10)
    centers_all = R.get_mask_voxels(roi)
    assert (rawseg[~roi] == 0).all() and (prob[~roi] == 0).all()            # nothing written outside the ROI
    rng = np.random.RandomState(0)
    where = np.array([smap[c] for c in centers_all])
    pick = set(rng.choice(len(centers_all), 700, replace=False).tolist())
    for k in range(1, 15):
        idx = np.flatnonzero(where == k)
        pick.update(rng.choice(idx, min(70, len(idx)), replace=False).tolist())
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
    # per-class Dice between the GPU labels and the CPU oracle labels on those voxels: exactly 1, over >= 12 classes
    seen = 0
    for c in range(15):
        a, b = rawseg[x, y, z][clear] == c, p.argmax(1)[clear] == c
        if a.sum() + b.sum():
            seen += 1
            assert 2.0 * (a & b).sum() / (a.sum() + b.sum()) == 1.0
    assert seen >= 12, 'only %d classes among the compared voxels' % seen
    # post-processing of the GPU labels equals the literal restatement (quirk included), with most structures live
    assert np.array_equal(seg, R.post_process_segmentation(rawseg, mask.astype(bool)))


def test_cohort_on_a_cold_net_with_many_reader_threads(tmp_path):
    """test_cohort's reader threads call net.normalize_volume / net.roi_centers, each of which needs the libscs handle: on a Net
    that has never been used (build_model + load_params_from, then straight into test_cohort) several threads reach the lazy
    handle creation at once.  It is created once, under a lock, with the parameters set - the files equal those of test_scan
    on a warm net."""
    import gzip
    a, b = str(tmp_path / 'a'), str(tmp_path / 'b')
    for i in range(6):
        make_subject(os.path.join(a, 's%d' % i), shape=(30, 28, 26), seed=60 + i, steered=True)
        make_subject(os.path.join(b, 's%d' % i), shape=(30, 28, 26), seed=60 + i, steered=True)
    opts = options(a, test_batch_size=100000)
    warm = build_model(os.path.join(PKG, 'nets'), opts)
    for i in range(6):
        base.test_scan(warm, os.path.join(a, 's%d' % i, 'T1.nii.gz'), opts)
    cold = build_model(os.path.join(PKG, 'nets'), options(b, test_batch_size=100000))
    assert cold._handle is None
    res = base.test_cohort(cold, [os.path.join(b, 's%d' % i, 'T1.nii.gz') for i in range(6)], options(b, test_batch_size=100000),
                           shard=False, readers=6, writers=2)
    assert len(res) == 6
    for i in range(6):
        for f in ('out_subcortical_prob.nii.gz', 'out_subcortical_seg_prec.nii.gz'):
            with gzip.open(os.path.join(a, 's%d' % i, f)) as fa, gzip.open(os.path.join(b, 's%d' % i, f)) as fb:
                assert fa.read() == fb.read(), (i, f)


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


def _roi_case(case):
    from scipy import ndimage
    from cnn_cort import synth
    rng = np.random.RandomState(31)
    if case == 'blob':                  # a dilated blob well inside the volume: box extents 45 x 41 x 38 (not multiples of anything)
        shape = (72, 64, 60)
        m = np.zeros(shape, bool)
        m[20:45, 18:39, 16:34] = rng.rand(25, 21, 18) > 0.97
        roi = ndimage.binary_dilation(m, iterations=10)
    elif case == 'border':              # an ROI that runs into three faces of the volume: patches hang over the edge (zero padding)
        shape = (40, 44, 36)
        roi = np.zeros(shape, bool)
        roi[0:19, 30:44, 0:23] = True
        roi[5:9, 33:36, 4:9] = False    # with a hole
    elif case == 'thin':                # one voxel thick along z, 5 along y: degenerate boxes in two views
        shape = (48, 40, 33)
        roi = np.zeros(shape, bool)
        roi[7:40, 17:22, 16] = True
    else:                               # scattered voxels: the box is mostly empty
        shape = (60, 60, 60)
        roi = rng.rand(*shape) > 0.97
        roi[:3] = roi[-3:] = False
    vol = base._zscore(synth.smooth_volume(shape, sigma=2.0, seed=32, as_int=True), False).astype(np.float32)
    centers = np.argwhere(roi)
    atlas = rng.dirichlet(np.ones(15) * 0.3, len(centers)).astype(np.float32)
    return vol, centers, atlas


@pytest.mark.parametrize('case', ['blob', 'border', 'thin', 'scattered'])
def test_shared_tower_evaluation_gives_the_bits_of_the_patch_path(weights, case):
    """scs_forward_volume_shared (csrc/dense_eval.h: per-slice maps computed tile by tile with the patch kernels, 16 pooling phases,
    3x3 windows gathered per voxel) against scs_forward_volume_rows (one tower evaluation per voxel and view): the same kernels
    see the same numbers in the same order, so labels AND probabilities must be equal bit for bit - inside the volume, over its
    border, for degenerate boxes and for a sparse centre list - with host and with device-resident inputs."""
    import torch
    vol, centers, atlas = _roi_case(case)
    net = exact_f32_form(build_model(os.path.join(PKG, 'nets'), options('/nonexistent')))
    lp, pp = net.predict_volume(vol, centers, atlas_rows=atlas, want_probs=True, batch=10 ** 9, towers='patch')
    assert net.last_scan_towers == 'patch'
    ls, ps = net.predict_volume(vol, centers, atlas_rows=atlas, want_probs=True, towers='shared')
    assert net.last_scan_towers == 'shared'
    print('%s: %d centres in a box of %d voxels' % (case, len(centers), np.prod(centers.max(0) - centers.min(0) + 1)))
    assert np.array_equal(ls, lp)
    assert np.array_equal(ps, pp), 'max difference %g' % np.abs(ps - pp).max()
    # device-resident volume and centres, labels only
    ld, _ = net.predict_volume(torch.from_numpy(vol).cuda(), torch.from_numpy(np.ascontiguousarray(centers, dtype=np.int32)).cuda(), atlas_rows=atlas,
                               towers='shared')
    assert np.array_equal(ld, lp)
    # 'auto' picks by density
    net.predict_volume(vol, centers, atlas_rows=atlas, towers='auto')
    assert net.last_scan_towers == ('shared' if case in ('blob', 'border') else 'patch')
    with pytest.raises(ValueError):
        net.predict_volume(vol, centers, atlas4d=np.zeros(vol.shape + (15,), np.float32), towers='shared')


def test_shared_tower_entry_refuses_a_centre_outside_its_box(handle):
    """scs_forward_volume_shared through the C ABI with a box that does not contain every centre: the tile maps cover the box only,
    so the call counts such centres on the device and fails with SCS_ERR_ARG instead of returning stale rows (include/scs.h);
    the same call with the right box succeeds."""
    import torch
    from cnn_cort import _hip
    vol, centers, atlas = _roi_case('blob')
    n = len(centers)
    d_vol = torch.from_numpy(vol).cuda()
    d_cen = torch.from_numpy(np.ascontiguousarray(centers, dtype=np.int32)).cuda()
    d_atl = torch.from_numpy(atlas).cuda()
    d_lab = torch.full((n,), -7, dtype=torch.int32, device='cuda:0')
    lo, hi = centers.min(0), centers.max(0)
    box = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
    st = torch.cuda.current_stream().cuda_stream
    handle.forward_volume_shared(d_vol, vol.shape, d_cen, d_atl, n, box, None, d_lab, st)
    good = d_lab.cpu().numpy().copy()
    assert (good >= 0).all()
    for short in ((lo[0] + 1, hi[0], lo[1], hi[1], lo[2], hi[2]), (lo[0], hi[0], lo[1], hi[1] - 2, lo[2], hi[2]), (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2] - 1)):
        with pytest.raises(_hip.ScsError, match='outside the box'):
            handle.forward_volume_shared(d_vol, vol.shape, d_cen, d_atl, n, short, None, d_lab, st)
    handle.forward_volume_shared(d_vol, vol.shape, d_cen, d_atl, n, box, None, d_lab, st)      # the handle is fine afterwards
    assert np.array_equal(d_lab.cpu().numpy(), good)


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


@pytest.mark.parametrize('dtype,order', [(np.int16, 'F'), (np.float32, 'C'), (np.uint8, 'F'), (np.float64, 'F')])
def test_device_zscore_equals_the_numpy_normalisation(dtype, order):
    """Net.normalize_volume (scs_zscore_volume: raw voxels in file order -> float32 [X,Y,Z]) against the reference's
    `(image - image[nz].mean()) / image[nz].std()` (base.py:357-358) followed by the float32 cast: equal to one float32 ulp (the
    mean / std sums run in a different order), for the element types and memory orders a T1 file comes in."""
    from cnn_cort import synth
    from cnn_cort.nets import Net
    shape = (37, 41, 29)
    vol = synth.smooth_volume(shape, sigma=3.0, seed=5)
    vol = (np.round(vol) if np.dtype(dtype).kind in 'iu' else vol * 1.37).astype(dtype)
    if dtype == np.uint8:
        vol = (synth.smooth_volume(shape, sigma=3.0, seed=5) / 4).astype(np.uint8)
    vol = np.asfortranarray(vol) if order == 'F' else np.ascontiguousarray(vol)
    net = Net(device=0)
    got = net.normalize_volume(vol)
    assert tuple(got.shape) == shape and got.is_contiguous()
    want = R.zscore_inference(vol.astype(np.float64) if np.dtype(dtype).kind == 'f' else vol).astype(np.float32)
    g = got.cpu().numpy()
    assert np.abs(g - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    assert np.mean(g != want) < 1e-3                                   # all but a few roundings identical
    if dtype == np.float32:
        # A float32 T1 goes through the reference's formula ENTIRELY in float32 (NumPy's pairwise float32 mean / std, float32
        # subtraction and division: base.py:358 with a float32 image).  The device computes mean / std in double (two passes)
        # and rounds once: a stated deviation, bounded here against the unmodified formula -
        # a few float32 ulps of the largest value.
        nz = vol[np.nonzero(vol)]
        ref32 = (vol - nz.mean()) / nz.std()
        assert ref32.dtype == np.float32
        assert np.abs(g - ref32).max() <= 4e-6 * max(1.0, np.abs(ref32).max())


def test_device_zscore_keeps_its_digits_when_the_mean_dwarfs_the_spread():
    """mean 30000, spread 3 (a float32 volume whose E[x^2] - mean^2 has no digits left in float64 either way worth trusting at
    float32 output precision only if formed carefully): the two-pass variance agrees with NumPy's."""
    from cnn_cort.nets import Net
    rng = np.random.RandomState(2)
    vol = (30000.0 + 3.0 * rng.standard_normal((33, 31, 29))).astype(np.float32)
    vol[:4] = 0
    got = Net(device=0).normalize_volume(vol).cpu().numpy()
    want = R.zscore_inference(vol.astype(np.float64)).astype(np.float32)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.parametrize('shape,iters,order', [((23, 17, 31), 10, 'F'), ((40, 44, 36), 10, 'C'), ((9, 64, 5), 3, 'F'), ((33, 20, 18), 0, 'F')])
def test_device_roi_voxel_list_equals_dilation_plus_nonzero(shape, iters, order):
    """Net.roi_centers (scs_roi_prepare + scs_roi_centers) == get_mask_voxels(binary_dilation(mask, iterations)) row for row:
    same voxels, same (C) order as np.nonzero (base.py:369-370, 310-331), from float and byte masks in either memory order."""
    from scipy import ndimage
    from cnn_cort.nets import Net
    rng = np.random.RandomState(sum(shape) + iters)
    mask = rng.rand(*shape) < 0.004
    mask[0, 0, 0] = mask[-1, -1, -1] = True
    raw = (mask.astype(np.float32) * 2.5) if order == 'F' else mask.astype(np.uint8)
    raw = np.asfortranarray(raw) if order == 'F' else np.ascontiguousarray(raw)
    net = Net(device=0)
    centers, d_cen = net.roi_centers(raw, iters)
    want = base.get_mask_voxels(ndimage.binary_dilation(mask, iterations=iters) if iters else mask)
    assert centers.dtype == np.int64 and np.array_equal(centers, want)
    assert np.array_equal(d_cen.cpu().numpy(), want.astype(np.int32))
    empty, d_empty = net.roi_centers(np.zeros(shape, np.uint8), 2)
    assert empty.shape == (0, 3) and d_empty.shape[0] == 0


@pytest.mark.timeout(900)
def test_full_size_subject_of_baseline_config_4(tmp_path, weights):
    """BASELINE configs[4] at its stated size: one 256x256x192 int16 subject with a 15-channel prior and a registered mask whose
    10-voxel dilation gives a ~3.3e5-voxel ROI, out_probabilities and post_process on, through test_scan (raw volumes to the
    device, z-score + ROI voxel list there, gather-fused forward, files written).  Checked: the voxel list against scipy + NumPy;
    the probabilities of 2 000 ROI voxels bit for bit against the PATCH path (literal strided-view get_patches from the same
    normalised volume -> scs_forward_host); 16 of them against the float64 oracle on patches cut by the literal per-voxel
    procedure from the NumPy-normalised volume (1e-4); labels = argmax; nothing written outside the ROI."""
    import sys
    from scipy import ndimage
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    scan = bench.make_cohort(str(tmp_path), 1, seed=3, steered=True)[0]
    sub = os.path.dirname(scan)
    opts = options(str(tmp_path), test_batch_size=100000)
    net = exact_f32_form(build_model(os.path.join(PKG, 'nets'), opts))      # shared-tower maps against per-voxel towers, bit for bit
    minutes = base.test_scan(net, scan, opts)
    assert 0 <= minutes < 1.0
    raw = nifti.load(scan).get_data()
    mask = nifti.load(os.path.join(sub, 'tmp', 'MNI_subcortical_mask.nii.gz')).get_data()
    roi = ndimage.binary_dilation(mask, iterations=10)
    centers = base.get_mask_voxels(roi)
    assert 250000 < len(centers) < 450000 and net.last_scan_voxels == len(centers)
    prob = nifti.load(os.path.join(sub, 'out_subcortical_prob.nii.gz')).get_data()
    assert prob.shape == raw.shape + (15,) and prob.dtype == np.float64
    assert np.count_nonzero(prob.sum(-1)) == len(centers) and (prob[~roi] == 0).all()      # every ROI voxel written, nothing else
    pick = np.sort(np.random.RandomState(1).choice(len(centers), 2000, replace=False))
    c = centers[pick]
    got = prob[c[:, 0], c[:, 1], c[:, 2]]
    # patch path on the same normalised volume: bit-identical
    vol = net.normalize_volume(raw).cpu().numpy()
    atlas_rows = base.atlas_rows_from_file(os.path.join(sub, 'tmp', 'MNI_sub_probabilities.nii.gz'), c)
    X = {'in%d' % (i + 1): base.get_patches(vol, c, (32, 32), m)[:, None] for i, m in enumerate(('axial', 'coronal', 'saggital'))}
    X['in4'] = atlas_rows
    p_patch = net.predict_proba(X)
    # (2 000 rows go through the split-K form of the dense head, the subject's 32 512-row chunks through the tiled one: the same
    # products summed in a different order)
    assert np.abs(got - p_patch).max() < 5e-5       # (logits of +-80 on this synthetic subject: 1e-7 relative is 1e-5 of a probability)
    # ... and bit for bit where the batch is the same: the first internal chunk of the subject (scs_api.hip: balanced_chunk)
    k = -(-len(centers) // 32768)
    step = -(-(-(-len(centers) // k)) // 16) * 16
    c1 = centers[:step]
    X1 = {'in%d' % (i + 1): base.get_patches(vol, c1, (32, 32), m)[:, None] for i, m in enumerate(('axial', 'coronal', 'saggital'))}
    X1['in4'] = base.atlas_rows_from_file(os.path.join(sub, 'tmp', 'MNI_sub_probabilities.nii.gz'), c1)
    assert np.array_equal(prob[c1[:, 0], c1[:, 1], c1[:, 2]], net.predict_proba(X1).astype(np.float64))
    # float64 oracle on literally cut patches of the NumPy-normalised volume
    vn = R.zscore_inference(raw)
    sel = pick[::125][:16]
    cs = [tuple(int(v) for v in centers[i]) for i in sel]
    views = [np.stack([np.array(R.get_patches(vn, cs, (32, 32), mode=m)).astype(np.float32)], axis=1) for m in ('axial', 'coronal', 'saggital')]
    atlas = nifti.load(os.path.join(sub, 'tmp', 'MNI_sub_probabilities.nii.gz')).get_data()
    p_ref = O.forward(weights, views[0], views[1], views[2], R.atlas_vectors(atlas, cs))
    g16 = prob[tuple(np.array(cs).T)]
    assert np.abs(g16 - p_ref).max() < 1e-4
    # labels: the post-processed file exists with the T1's geometry; the raw labels of a second call are argmax of the probabilities
    seg = nifti.load(os.path.join(sub, 'out_subcortical_seg_prec.nii.gz')).get_data()
    assert seg.shape == raw.shape and seg.dtype == raw.dtype
    base.test_scan(net, scan, options(str(tmp_path), test_batch_size=100000, post_process='False', out_probabilities='False'))
    rawseg = nifti.load(os.path.join(sub, 'out_subcortical_rawseg.nii.gz')).get_data()
    assert np.array_equal(rawseg[c[:, 0], c[:, 1], c[:, 2]], got.argmax(1)) and (rawseg[~roi] == 0).all()
    live = np.unique(rawseg)
    assert len(live) >= 13, 'classes at volume level: %s' % live           # the steered prior: (nearly) every structure is present
    # post-processing against the LITERAL procedure (base.py:460-480 restated in oracle/patches_ref.py) on the ROI's bounding
    # box: every labelled voxel lies inside the box, component numbering follows the raster order of first voxels (which a crop
    # preserves), so the literal result on the crop is the full result restricted to it; outside the box every voxel is
    # background of every class and takes the value a background voxel of the box takes
    box = tuple(slice(int(lo), int(hi) + 1) for lo, hi in zip(centers.min(0), centers.max(0)))
    lit = R.post_process_segmentation(rawseg[box], mask[box].astype(bool))
    assert np.array_equal(seg[box], lit)
    bg_in_box = lit[(rawseg[box] == 0)]
    outside = np.ones(raw.shape, bool)
    outside[box] = False
    assert bg_in_box.size and (bg_in_box == bg_in_box.flat[0]).all() and (seg[outside] == bg_in_box.flat[0]).all()
