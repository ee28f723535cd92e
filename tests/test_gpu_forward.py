"""Parity of the HIP forward path (through the C ABI) with the float64 oracle - needs an MI355X.

Tolerances (BASELINE.json north_star): softmax probabilities within 1e-4 absolute of the oracle; argmax labels
identical wherever the oracle's own top-2 margin exceeds float32 noise (1e-5) and, for closer calls, one of the
oracle's top two classes."""
import numpy as np
import pytest
import torch

from conftest import gold
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-4


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0', dtype=dtype)


def run_forward(handle, ax, co, sa, at, want_probs=True):
    n = ax.shape[0]
    probs = torch.full((n, 15), -7.0, device='cuda:0') if want_probs else None
    labels = torch.full((n,), -7, dtype=torch.int32, device='cuda:0')
    handle.forward(dev(ax), dev(co), dev(sa), dev(at), n, probs, labels, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return (probs.cpu().numpy() if want_probs else None), labels.cpu().numpy()


def check_labels(labels, ref_probs):
    srt = np.sort(ref_probs, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    ref = ref_probs.argmax(1)
    clear = margin > 1e-5
    assert np.array_equal(labels[clear], ref[clear])
    top2 = np.argsort(ref_probs, axis=1)[:, -2:]
    assert all(l in t for l, t in zip(labels[~clear], top2[~clear]))


@pytest.mark.parametrize('name', ['g1_random.npz', 'g2_volume.npz', 'g3_classes.npz', 'g4_ties.npz'])
def test_golden_sets(handle, name):
    g = gold(name)
    probs, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
    assert np.isfinite(probs).all()
    err = np.abs(probs - g['probs']).max()
    assert err < PROB_TOL, 'max |p - p_oracle| = %g' % err
    assert np.abs(probs.sum(1) - 1).max() < 1e-5
    check_labels(labels, g['probs'])
    # labels are the argmax of the returned float32 probabilities (nolearn predict = argmax(predict_proba))
    assert np.array_equal(labels, probs.argmax(1))
    # The probabilities of g1..g3 are saturated (top-2 margins >= 0.88): a 1e-4 bound on them passes almost any logit error.
    # The logits themselves (out_layer pre-softmax, nets.py:231) are compared at 2e-5 of their scale - the per-layer bar.
    n = g['ax'].shape[0]
    logits = handle.get_activation(7, n * 16).reshape(n, 16)[:, :15]
    scale = max(1.0, float(np.abs(g['logits']).max()))
    lerr = float(np.abs(logits - g['logits']).max())
    assert lerr <= 2e-5 * scale, 'max |logit - logit_oracle| = %g (scale %g)' % (lerr, scale)
    # near-tie rows (oracle top-2 margin <= 1e-5, where check_labels accepts either of the top two): none in g1..g3
    srt = np.sort(g['probs'], axis=1)
    relaxed = int(((srt[:, -1] - srt[:, -2]) <= 1e-5).sum())
    print('%s: %d of %d rows in the relaxed near-tie band; max logit error %.3g of scale %.3g' % (name, relaxed, n, lerr, scale))
    if name != 'g4_ties.npz':
        assert relaxed == 0


def test_per_layer_activations(handle, weights):
    g = gold('g2_volume.npz')
    n = 24
    args = [g[k][:n] for k in ('ax', 'co', 'sa', 'atlas')]
    tr = {}
    O.forward(weights, *args, trace=tr)
    run_forward(handle, *args)
    towers = ('axial', 'coronal', 'saggital')

    def close(got, ref, what):
        scale = max(1.0, np.abs(ref).max())
        err = np.abs(got - ref).max()
        assert err <= 2e-5 * scale, '%s: max err %g (scale %g)' % (what, err, scale)

    for which, key, per in ((0, 'pool1', 20 * 196), (1, 'conv3', 40 * 144), (2, 'pool2', 40 * 25), (3, 'conv5', 540)):
        got = handle.get_activation(which, 3 * n * per).reshape(3, n, -1)
        for t, tn in enumerate(towers):
            close(got[t], tr['%s_%s' % (tn, key)].reshape(n, -1), '%s_%s' % (tn, key))
    feat = handle.get_activation(4, n * 540).reshape(n, 540)
    close(feat, np.concatenate([tr['%s_d1' % t] for t in towers], axis=1), 'd1 concat')
    fc1 = handle.get_activation(5, n * 556).reshape(n, 556)
    close(fc1[:, :540], tr['FC1'], 'FC1')
    if handle.get_option('fwd_split') > 0:     # (forced on from the environment: the prior columns travel as two f16 pieces, 22 bits)
        close(fc1[:, 540:555], args[3], 'prior columns')
    else:
        assert np.array_equal(fc1[:, 540:555], args[3])
    assert np.all(fc1[:, 555] == 0)
    fc2 = handle.get_activation(6, n * 272).reshape(n, 272)
    close(fc2[:, :270], tr['fc_2'], 'fc_2')
    logits = handle.get_activation(7, n * 16).reshape(n, 16)
    close(logits[:, :15], tr['logits'], 'logits')


@pytest.mark.parametrize('name', ['g1_random.npz', 'g2_volume.npz', 'g3_classes.npz', 'g4_ties.npz'])
def test_fused_conv3_to_conv5_kernel_gives_the_bits_of_the_three_launches(handle, name):
    """conv3 -> conv4 + pool2 -> conv5 as one kernel per patch-tower (csrc/conv_fused.h, option fwd_fused = 1: the two
    activations in between never leave LDS) against the three launches (the default): the same MFMA sequences per output, so
    probabilities, labels, logits and the block-5 activations are equal bit for bit - and the debug form (fwd_fused = 2) stores
    the same conv3 / pool2 activations the three launches leave in memory."""
    g = gold(name)
    n = g['ax'].shape[0]
    got = {}
    for form in (0, 1, 2):
        handle.set_option('fwd_fused', form)
        try:
            probs, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
            acts = {w: handle.get_activation(w, 3 * n * per) for w, per in ((3, 540),) + (((1, 5760), (2, 1000)) if form != 1 else ())}
            logits = handle.get_activation(7, n * 16)
            if form == 1:
                with pytest.raises(Exception):
                    handle.get_activation(1, 3 * n * 5760)
        finally:
            handle.set_option('fwd_fused', -1)
        got[form] = (probs, labels, logits, acts)
    for form in (1, 2):
        assert np.array_equal(got[form][0], got[0][0]) and np.array_equal(got[form][1], got[0][1]) and np.array_equal(got[form][2], got[0][2])
        assert np.array_equal(got[form][3][3], got[0][3][3])
    assert np.array_equal(got[2][3][1], got[0][3][1]) and np.array_equal(got[2][3][2], got[0][3][2])


@pytest.mark.parametrize('n', [1, 3, 4, 5, 15, 16, 17, 63, 130])
def test_ragged_batch_sizes(handle, n):
    """Partial patch groups (conv3/4: 4 per workgroup, conv5: 16, GEMM: 128 rows) must not leak or drop rows."""
    g = gold('g2_volume.npz')
    args = [g[k][:n] for k in ('ax', 'co', 'sa', 'atlas')]
    probs, labels = run_forward(handle, *args)
    assert np.abs(probs - g['probs'][:n]).max() < PROB_TOL
    check_labels(labels, g['probs'][:n])


def test_chunking_is_invisible(handle):
    g = gold('g2_volume.npz')
    args = [g[k] for k in ('ax', 'co', 'sa', 'atlas')]
    full, lab_full = run_forward(handle, *args)
    old = handle.set_chunk(0)
    try:
        assert handle.set_chunk(48) == 48
        part, lab_part = run_forward(handle, *args)          # 256 patches -> 5 full chunks + 16
    finally:
        handle.set_chunk(old)
    assert np.array_equal(full, part) and np.array_equal(lab_full, lab_part)


def test_empty_and_optional_outputs(handle):
    g = gold('g1_random.npz')
    handle.forward(None, None, None, None, 0, None, None, 0)                      # n = 0 is a no-op
    _, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'], want_probs=False)
    check_labels(labels, g['probs'])


def test_host_pointer_entry_matches_device_entry(handle):
    g = gold('g1_random.npz')
    probs, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
    n = g['ax'].shape[0]
    hp = np.empty((n, 15), np.float32)
    hl = np.empty((n,), np.int32)
    handle.forward_host(g['ax'], g['co'], g['sa'], g['atlas'], n, hp, hl)
    assert np.array_equal(hp, probs) and np.array_equal(hl, labels)
    # several chunks: the upload of chunk i+1 overlaps the kernels of chunk i (double-buffered staging)
    old = handle.set_chunk(0)
    try:
        handle.set_chunk(48)
        hp2 = np.full((n, 15), -1, np.float32)
        hl2 = np.full((n,), -1, np.int32)
        handle.forward_host(g['ax'], g['co'], g['sa'], g['atlas'], n, hp2, hl2)
        handle.forward_host(g['ax'], g['co'], g['sa'], g['atlas'], n, None, hl2)     # labels only
    finally:
        handle.set_chunk(old)
    assert np.array_equal(hp2, probs) and np.array_equal(hl2, labels)


def test_gather_fused_volume_path_equals_patch_path(handle):
    """scs_forward_volume cuts the same patches the literal get_patches restatement produced for g2 (border
    centres included) and applies the zero-row atlas rule: results must equal the patch entry bit for bit."""
    g = gold('g2_volume.npz')
    from oracle import patches_ref as R
    vol = R.zscore_inference(g['volume']).astype(np.float32)
    probs, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
    n = g['centers'].shape[0]
    d_pr = torch.empty((n, 15), device='cuda:0')
    d_lb = torch.empty((n,), dtype=torch.int32, device='cuda:0')
    handle.forward_volume(dev(vol), vol.shape, dev(g['centers'], torch.int32), dev(g['atlas4d']), n, d_pr, d_lb,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_pr.cpu().numpy(), probs)
    assert np.array_equal(d_lb.cpu().numpy(), labels)


def test_full_size_properties(handle):
    """BASELINE config 2 size (100 000 patches): size-independent properties - rows sum to 1, permutation
    equivariance (a shuffled batch gives the shuffled result bit for bit), determinism across calls."""
    from cnn_cort import synth
    n = 100000
    ax, co, sa, at = synth.patch_batch(n, seed=0)
    d = [dev(a) for a in (ax, co, sa, at)]
    st = torch.cuda.current_stream().cuda_stream
    p1 = torch.empty((n, 15), device='cuda:0')
    l1 = torch.empty((n,), dtype=torch.int32, device='cuda:0')
    handle.forward(*d, n, p1, l1, st)
    perm = torch.randperm(n, device='cuda:0', generator=torch.Generator(device='cuda:0').manual_seed(1))
    d2 = [a[perm].contiguous() for a in d]
    p2 = torch.empty_like(p1)
    l2 = torch.empty_like(l1)
    handle.forward(*d2, n, p2, l2, st)
    p3 = torch.empty_like(p1)
    handle.forward(*d, n, p3, None, st)
    torch.cuda.synchronize()
    assert torch.isfinite(p1).all()
    assert (p1.sum(1) - 1).abs().max().item() < 1e-5
    assert torch.equal(p1[perm], p2) and torch.equal(l1[perm], l2)
    assert torch.equal(p1, p3)
    assert torch.equal(l1.long(), p1.argmax(1))
    # spot-check a slice against the oracle
    from cnn_cort import params as P
    from conftest import WEIGHTS
    idx = np.arange(0, n, n // 16)[:16]
    ref = O.forward(P.load_params(WEIGHTS), ax[idx], co[idx], sa[idx], at[idx])
    assert np.abs(p1.cpu().numpy()[idx] - ref).max() < PROB_TOL


def test_error_codes_and_call_order(handle):
    """C-ABI error behaviour (include/scs.h): negative status + text, never an abort; the handle stays usable."""
    from cnn_cort import _hip
    lib = handle.lib
    g = gold('g1_random.npz')
    d = [dev(g[k][:4]) for k in ('ax', 'co', 'sa', 'atlas')]
    lab = torch.empty((4,), dtype=torch.int32, device='cuda:0')
    # call order: a fresh handle has no parameters
    h2 = _hip.Handle(0)
    try:
        rc = lib.scs_forward(h2.h, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 4, None, lab.data_ptr(), None)
        assert rc == -3 and b'scs_set_params' in lib.scs_last_error()
        with pytest.raises(_hip.ScsError):
            h2.get_params()
        # wrong arena size
        bad = np.zeros(100, np.float32)
        assert lib.scs_set_params(h2.h, bad.ctypes.data, bad.size) == -1
    finally:
        h2.close()
    # bad arguments on a good handle
    assert lib.scs_forward(handle.h, None, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 4, None, lab.data_ptr(), None) == -1
    assert lib.scs_forward(handle.h, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), -1, None, lab.data_ptr(), None) == -1
    assert lib.scs_forward(None, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 4, None, lab.data_ptr(), None) == -1
    assert lib.scs_forward_volume(handle.h, d[0].data_ptr(), 0, 4, 4, d[1].data_ptr(), d[2].data_ptr(), 4, None, lab.data_ptr(), None) == -1
    assert lib.scs_last_error() != b''
    # ... and it still computes
    _, labels = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'], want_probs=False)
    check_labels(labels, g['probs'])


def test_two_handles_on_two_streams(handle, weights):
    """The library is re-entrant per handle: two handles driven from two HIP streams (and two host threads) at the
    same time give the same bits as one handle alone."""
    import threading
    from cnn_cort import _hip, params as P
    g = gold('g2_volume.npz')
    ref_p, ref_l = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
    other = _hip.Handle(0)
    other.set_params(P.params_to_flat(weights))
    outs = {}

    def work(name, h):
        s = torch.cuda.Stream(device='cuda:0')
        with torch.cuda.stream(s):
            ins = [dev(g[k]) for k in ('ax', 'co', 'sa', 'atlas')]
            n = ins[0].shape[0]
            pr = torch.empty((n, 15), device='cuda:0')
            lb = torch.empty((n,), dtype=torch.int32, device='cuda:0')
            for _ in range(5):
                h.forward(ins[0], ins[1], ins[2], ins[3], n, pr, lb, s.cuda_stream)
            s.synchronize()
            outs[name] = (pr.cpu().numpy(), lb.cpu().numpy())

    try:
        th = [threading.Thread(target=work, args=('a', handle)), threading.Thread(target=work, args=('b', other))]
        [t.start() for t in th]
        [t.join() for t in th]
    finally:
        other.close()
    for name in ('a', 'b'):
        assert np.array_equal(outs[name][0], ref_p) and np.array_equal(outs[name][1], ref_l)


def test_forward_is_deterministic(handle):
    g = gold('g2_volume.npz')
    first = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
    for _ in range(4):
        again = run_forward(handle, g['ax'], g['co'], g['sa'], g['atlas'])
        assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])


def test_large_n_is_chunked_consistently(handle):
    """300 001 patches in one call (ten internal chunks, 3.7 GB of inputs): every row is a distribution, and slices
    from the beginning, the middle and the ragged tail equal separate small calls on the same rows."""
    n = 300001
    gen = torch.Generator(device='cuda:0')
    gen.manual_seed(1)
    ax, co, sa = [torch.randn((n, 1, 32, 32), generator=gen, device='cuda:0') for _ in range(3)]
    at = torch.rand((n, 15), generator=gen, device='cuda:0')
    at = at / at.sum(1, keepdim=True)
    pr = torch.empty((n, 15), device='cuda:0')
    lb = torch.empty((n,), dtype=torch.int32, device='cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    handle.forward(ax, co, sa, at, n, pr, lb, st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(pr).all()) and float((pr.sum(1) - 1).abs().max()) < 1e-5
    assert bool((pr.argmax(1).to(torch.int32) == lb).all())
    m = 555
    for lo in (0, 150000, n - m):
        p2 = torch.empty((m, 15), device='cuda:0')
        l2 = torch.empty((m,), dtype=torch.int32, device='cuda:0')
        handle.forward(ax[lo:lo + m], co[lo:lo + m], sa[lo:lo + m], at[lo:lo + m], m, p2, l2, st)
        torch.cuda.synchronize()
        assert float((p2 - pr[lo:lo + m]).abs().max()) < 1e-5      # the small call takes the split-K dense kernels
        srt = torch.sort(p2, dim=1).values
        clear = (srt[:, -1] - srt[:, -2]) > 1e-4
        assert bool((l2 == lb[lo:lo + m])[clear].all())


@pytest.mark.gpu
def test_mfma_issue_rate_probe_is_a_plausible_fraction_of_the_nominal_peak():
    """scs_probe_f32_mfma (the figure bench.py sets beside the nominal 157.3 TFLOP/s): back-to-back f32 MFMAs on every SIMD cannot
    exceed the nominal peak and on an MI355X do not fall below half of it."""
    from cnn_cort import _hip
    h = _hip.Handle(0)
    try:
        rate = h.probe_f32_mfma()
    finally:
        h.close()
    assert 0.5 * 157.3 < rate <= 157.3 * 1.02, rate
