import os, sys
import numpy as np
import torch
ROOT = '/root/repo'
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from cnn_cort import _hip, dist, params as P, synth
from oracle import np_oracle as O
W = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0', dtype=dt)
h = _hip.Handle(0)
st = torch.cuda.current_stream().cuda_stream
n = 64
for off, sc in ((0.0, 1.0), (3.0, 1.0), (10.0, 0.5), (50.0, 1.0)):
    h.set_params(P.params_to_flat(W))
    ax, co, sa, at = synth.patch_batch(n, seed=5)
    ax, co, sa = [(v * sc + off).astype(np.float32) for v in (ax, co, sa)]
    rng = np.random.RandomState(n)
    y = rng.randint(0, 15, n).astype(np.uint8)
    m = {'%s_l1drop' % t: (rng.rand(n, 60, 3, 3) < 0.5).astype(np.uint8) for t in P.TOWERS}
    m['f1_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8); m['f2_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    loss_ref, gref, bn_ref, _ = O.train_forward_backward(W, ax, co, sa, at, y, m)
    m5 = np.stack([m['%s_l1drop' % t].reshape(n, 540) for t in P.TOWERS])
    pm = np.concatenate([m5.ravel(), m['f1_drop'].ravel(), m['f2_drop'].ravel()]).astype(np.uint8)
    loss = h.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, masks=dev(pm, torch.uint8), stream=st)
    torch.cuda.synchronize()
    ptr, cnt = h.grad_arena()
    got = P.flat_to_params(dist.wrap_device_floats(ptr, cnt, 0).cpu().numpy().copy())
    newp = P.flat_to_params(h.get_params())
    run = O.bn_running_update(W, bn_ref)
    out = []
    for name in ('axial_ch_conv1', 'axial_ch_conv1_bn', 'axial_ch_conv2', 'axial_ch_conv2_bn'):
        for i in range(len(gref[name])):
            ref = np.asarray(gref[name][i]); s_ = max(np.abs(ref).max(), 1e-6)
            if P.is_trainable(name, i): out.append('%s[%d] %.2g' % (name.replace('axial_ch_', ''), i, np.abs(got[name][i] - ref).max() / s_))
    e_inv = np.abs(newp['axial_ch_conv1_bn'][3] - run['axial_ch_conv1_bn'][3]).max() / np.abs(np.asarray(run['axial_ch_conv1_bn'][3])).max()
    print('offset %g scale %g: loss %.6f ref %.6f | run inv_std err %.2g | %s' % (off, sc, loss, loss_ref, e_inv, ' '.join(out)), flush=True)
