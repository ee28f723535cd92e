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


def test_a_flagged_chunk_is_repeated_on_the_device_and_the_others_are_not(split):
    """The repeat is per internal chunk and happens on the device (guarded f32 launches behind the split kernels of the chunk, no host
    round trip): with 64-row chunks and out-of-range inputs in rows 64..127 only, that chunk carries the f32 path's bits, the
    other three the split kernels', and the device counter reads 1."""
    g = gold('g1_random.npz')
    n = 256
    rep = lambda a: np.concatenate([a] * (n // a.shape[0] + 1))[:n].copy()
    ax, co, sa, at = (rep(g[k]) for k in ('ax', 'co', 'sa', 'atlas'))
    assert split.set_chunk(64) == 64
    try:
        p_split, l_split = run_forward(split, ax, co, sa, at)                  # clean: all four chunks on the split kernels
        assert split.get_option('split_fallbacks') == 0
        ax[64:128] *= 3.0e4
        probs, labels = run_forward(split, ax, co, sa, at)
        assert split.get_option('split_fallbacks') == 1
        split.set_option('fwd_split', -1)
        p32, l32 = run_forward(split, ax, co, sa, at)
    finally:
        split.set_chunk(32768)
    hot = np.zeros(n, bool)
    hot[64:128] = True
    assert np.array_equal(probs[hot], p32[hot]) and np.array_equal(labels[hot], l32[hot])
    assert np.array_equal(probs[~hot], p_split[~hot]) and np.array_equal(labels[~hot], l_split[~hot])
    assert not np.array_equal(p_split[~hot], p32[~hot])                      # (the two forms do differ in the last bits)


def test_the_split_entries_enqueue_and_return(split):
    """work is ENQUEUED on the caller's stream.  With fwd_split the call used to wait for the stream to read the overflow
    flag; now nothing in it synchronises - shown by capturing it into a HIP graph (any synchronisation or blocking copy inside a
    capture is an error) and replaying the graph on new inputs, clean and overflowing."""
    from test_gpu_forward import dev
    g = gold('g1_random.npz')
    n = g['ax'].shape[0]
    d = [dev(g[k]) for k in ('ax', 'co', 'sa', 'atlas')]
    pr = torch.empty((n, 15), device='cuda:0')
    lb = torch.empty((n,), dtype=torch.int32, device='cuda:0')
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        split.forward(*d, n, pr, lb, s.cuda_stream)                            # warm-up: the workspace exists before the capture
    s.synchronize()
    want = pr.clone()
    graph = torch.cuda.CUDAGraph()
    pr.zero_()
    with torch.cuda.graph(graph, stream=s):
        split.forward(*d, n, pr, lb, torch.cuda.current_stream().cuda_stream)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(pr, want) and split.get_option('split_fallbacks') == 0
    for t in d[:3]:
        t.mul_(3.0e4)                                                          # same buffers, out-of-range values: the replay repeats on f32
    graph.replay()
    torch.cuda.synchronize()
    assert split.get_option('split_fallbacks') == 1
    split.set_option('fwd_split', -1)
    p32 = torch.empty_like(pr)
    split.forward(*d, n, p32, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(pr, p32)


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
    assert torch.equal(l1.long(), p1.argmax(1))
    assert split.get_option('split_fallbacks') == 0
    # spot-check a slice against the float64 oracle (as the f32 twin does: tests/test_gpu_forward.py::test_full_size_properties)
    from cnn_cort import params as P
    from conftest import WEIGHTS
    idx = np.arange(0, n, n // 16)[:16]
    ref = O.forward(P.load_params(WEIGHTS), ax[idx], co[idx], sa[idx], at[idx])
    assert np.abs(p1.cpu().numpy()[idx] - ref).max() < PROB_TOL


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


@pytest.mark.timeout(900)
def test_whole_subject_through_test_scan_on_the_split_kernels(tmp_path, weights):
    """test_scan (base.py:401-458) with options['fwd_split'] = 'True' and the per-voxel towers (options['towers'] = 'patch') on the
    256x256x192 15-class subject of BASELINE configs[4]: the raw labels equal the f32 path's on every ROI voxel, per-class Dice 1.0 over
    all 15 classes, probabilities within 2e-5 of the f32 path's, and NO chunk fell back to the f32 kernels (split_fallbacks == 0: what
    is compared is the split kernels' output).  The default towers ('auto': shared-tower evaluation on a dense ROI) stay on the f32
    kernels and say so once."""
    import os
    import sys
    import warnings
    from conftest import ROOT, PKG
    sys.path.insert(0, ROOT)
    import bench
    from cnn_cort import base, nifti, nets
    from cnn_cort.nets import build_model
    scan = bench.make_cohort(str(tmp_path), 1, seed=3, steered=True)[0]
    sub = os.path.dirname(scan)
    opts = {'experiment': 'miccai2012_v1', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 20, 'patch_size': [32, 32],
            'load_weights': 'True', 'batch_size': 128, 'device': 0, 'crop': 'True', 'test_batch_size': 100000, 'out_probabilities': 'True',
            'post_process': 'False', 'debug': 'False', 'towers': 'patch'}
    out = {}
    for form in ('f32', 'split'):
        o = dict(opts, fwd_split='True') if form == 'split' else dict(opts)
        net = build_model(os.path.join(PKG, 'nets'), o)
        base.test_scan(net, scan, o)
        assert net.last_scan_towers == 'patch'
        seg = nifti.load(os.path.join(sub, 'out_subcortical_rawseg.nii.gz')).get_data().copy()
        prob = nifti.load(os.path.join(sub, 'out_subcortical_prob.nii.gz')).get_data()
        roi = prob.sum(-1) != 0
        out[form] = (seg, prob[roi].astype(np.float32), roi, net._handle.get_option('split_fallbacks'), net.last_scan_voxels)
        if form == 'split':
            assert net._handle.get_option('fwd_split') == 1
            nets.Net._warned_split_shared = False
            with pytest.warns(RuntimeWarning, match='shared-tower evaluation stays on the f32 kernels'):
                base.test_scan(net, scan, dict(o, towers='auto', out_probabilities='False'))
            assert net.last_scan_towers == 'shared'
            assert np.array_equal(nifti.load(os.path.join(sub, 'out_subcortical_rawseg.nii.gz')).get_data(), out['f32'][0])
            with warnings.catch_warnings():
                warnings.simplefilter('error')                                  # ... once per process
                base.test_scan(net, scan, dict(o, towers='auto', out_probabilities='False'))
    (s32, p32, r32, _, n32), (ssp, psp, rsp, fallbacks, nsp) = out['f32'], out['split']
    assert fallbacks == 0, '%d chunk(s) of the subject fell back to the f32 kernels' % fallbacks
    assert n32 == nsp > 250000 and np.array_equal(r32, rsp)
    assert np.array_equal(ssp, s32), '%d labels differ' % int((ssp != s32).sum())
    live = np.unique(s32)
    assert len(live) == 15, live
    for k in live:                                                               # per-class Dice (trivially 1.0 with equal labels: stated for the record)
        a, b = ssp == k, s32 == k
        assert 2.0 * (a & b).sum() / (a.sum() + b.sum()) == 1.0
    diff = np.abs(psp - p32).max(1)
    err = float(diff.max())
    # the logits of this steered subject reach +-80, so 1e-7 relative is 1e-5 of a probability near a tie (the f32 path itself is
    # 4.5e-5 from the float32 CPU port on it, bench.py `dice`): the two device paths are compared at 2.5e-4, and BOTH are held to the
    # 1e-4 bar against the float64 oracle on the 16 voxels where they differ most (patches cut by the literal per-voxel procedure)
    from oracle import patches_ref as R
    worst = np.argsort(diff)[-16:]
    cs = [tuple(int(v) for v in c) for c in np.argwhere(r32)[worst]]
    raw = nifti.load(scan).get_data()
    vn = R.zscore_inference(raw)
    views = [np.stack([np.array(R.get_patches(vn, cs, (32, 32), mode=m)).astype(np.float32)], axis=1) for m in ('axial', 'coronal', 'saggital')]
    atlas = nifti.load(os.path.join(sub, 'tmp', 'MNI_sub_probabilities.nii.gz')).get_data()
    p_ref = O.forward(weights, views[0], views[1], views[2], R.atlas_vectors(atlas, cs))
    e32, esp = float(np.abs(p32[worst] - p_ref).max()), float(np.abs(psp[worst] - p_ref).max())
    print('whole subject on the split kernels: %d ROI voxels, 15 classes, labels equal, fallbacks %d; max |p_split - p_f32| = %.3g; on the 16 voxels '
          'where the two paths differ most, against the float64 oracle: f32 path %.3g, split path %.3g' % (nsp, fallbacks, err, e32, esp))
    assert err < 2.5e-4
    assert e32 < 1e-4 and esp < 1e-4
