"""Per-array relative gradient error of scs_train_step against the float64 oracle at a few batch sizes (diagnostic)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from cnn_cort import _hip, dist, params as P, synth
from oracle import np_oracle as O

W = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0', dtype=dt)
h = _hip.Handle(0)
st = torch.cuda.current_stream().cuda_stream
for n in [int(x) for x in sys.argv[1:]] or [32, 37]:
    h.set_params(P.params_to_flat(W))
    ax, co, sa, at = synth.patch_batch(n, seed=100 + n)
    rng = np.random.RandomState(n)
    y = rng.randint(0, 15, n).astype(np.uint8)
    m = {'%s_l1drop' % t: (rng.rand(n, 60, 3, 3) < 0.5).astype(np.uint8) for t in P.TOWERS}
    m['f1_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8); m['f2_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    dbg = {}
    loss_ref, gref, bn_ref, _ = O.train_forward_backward(W, ax, co, sa, at, y, m, debug=dbg)
    m5 = np.stack([m['%s_l1drop' % t].reshape(n, 540) for t in P.TOWERS])
    pm = np.concatenate([m5.ravel(), m['f1_drop'].ravel(), m['f2_drop'].ravel()]).astype(np.uint8)
    loss = h.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, masks=dev(pm, torch.uint8), stream=st)
    torch.cuda.synchronize()
    ptr, cnt = h.grad_arena()
    got = P.flat_to_params(dist.wrap_device_floats(ptr, cnt, 0).cpu().numpy().copy())
    print('n = %d  loss %.6f ref %.6f' % (n, loss, loss_ref))
    if os.environ.get('DENSE_DBG'):
        dR2 = h.get_train_buffer(0, n * 272).reshape(n, 272)[:, :270].astype(np.float64)
        dcat = h.get_train_buffer(1, n * 540).reshape(n, 540).astype(np.float64)
        featd = h.get_train_buffer(3, n * 540).reshape(n, 540).astype(np.float64)
        R1 = h.get_train_buffer(4, n * 540).reshape(n, 540).astype(np.float64)
        W2 = np.asarray(W['fc_2'][0], np.float64)
        want = (dR2 @ W2[:540].T) * (2.0 * m['f2_drop']) * (R1 > 0)
        print('   dR2 vs oracle', np.abs(dR2 - dbg['dpre2']).max(), 'dcat vs oracle', np.abs(dcat - dbg['dpre1']).max(), 'R1 vs oracle', np.abs(R1 - dbg['r1']).max(), 'featd', np.abs(featd - dbg['cat_d']).max(), 'rows', np.nonzero(np.abs(dcat - dbg['dpre1']).max(1) > 1e-6)[0])
        print('   dcat vs formula', np.abs(dcat - want).max(), ' FC1 b vs colsum', np.abs(got['FC1'][1] - dcat.sum(0)).max(),
              ' oracle FC1 b vs colsum', np.abs(np.asarray(gref['FC1'][1]) - dcat.sum(0)).max(), ' fc2 b vs colsum dR2', np.abs(np.asarray(gref['fc_2'][1]) - dR2.sum(0)).max())
    for name, shapes in P.LAYER_SPEC:
        for i, _ in enumerate(shapes):
            ref = np.asarray(gref[name][i]); scale = max(np.abs(ref).max(), 1e-6)
            err = np.abs(got[name][i] - ref).max() / scale
            if err > 1e-4 and P.is_trainable(name, i):
                print('   %-24s[%d] rel err %.3g  scale %.3g' % (name, i, err, scale))
