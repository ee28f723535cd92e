import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from cnn_cort import _hip, dist, params as P, synth
W = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0', dtype=dt)
h = _hip.Handle(0)
st = torch.cuda.current_stream().cuda_stream
for n in [int(x) for x in sys.argv[1:]] or [32, 37]:
    h.set_params(P.params_to_flat(W))
    ax, co, sa, at = synth.patch_batch(n, seed=100 + n)
    rng = np.random.RandomState(n)
    y = rng.randint(0, 15, n).astype(np.uint8)
    m5 = (rng.rand(3, n, 540) < 0.5).astype(np.uint8); m1 = (rng.rand(n, 540) < 0.5).astype(np.uint8); m2 = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    pm = np.concatenate([m5.ravel(), m1.ravel(), m2.ravel()]).astype(np.uint8)
    h.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, masks=dev(pm, torch.uint8), stream=st)
    torch.cuda.synchronize()
    dR2 = h.get_train_buffer(0, n * 272).reshape(n, 272)[:, :270].astype(np.float64)
    dcat = h.get_train_buffer(1, n * 540).reshape(n, 540).astype(np.float64)
    dfeat = h.get_train_buffer(2, n * 540).reshape(n, 540).astype(np.float64)
    featd = h.get_train_buffer(3, n * 540).reshape(n, 540).astype(np.float64)
    R1 = h.get_train_buffer(4, n * 540).reshape(n, 540).astype(np.float64)
    W2 = np.asarray(W['fc_2'][0], np.float64)      # [555][270]
    W1 = np.asarray(W['FC1'][0], np.float64)       # [540][540]
    want = (dR2 @ W2[:540].T) * (2.0 * m2) * (R1 > 0)
    print('n', n, 'dcat max err', np.abs(dcat - want).max(), 'scale', np.abs(want).max(), 'rows bad', np.nonzero(np.abs(dcat - want).max(1) > 1e-6 * np.abs(want).max())[0][:20])
    ptr, cnt = h.grad_arena()
    got = P.flat_to_params(dist.wrap_device_floats(ptr, cnt, 0).cpu().numpy().copy())
    gW1 = featd.T @ dcat
    print('   FC1 W grad vs featd^T dcat(device):', np.abs(got['FC1'][0] - gW1).max(), 'scale', np.abs(gW1).max(), ' bias:', np.abs(got['FC1'][1] - dcat.sum(0)).max())
    gW1w = featd.T @ want
    print('   FC1 W grad vs featd^T dcat(want):  ', np.abs(got['FC1'][0] - gW1w).max())
