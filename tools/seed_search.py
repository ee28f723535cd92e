"""Which draw of the ragged-batch test has no ReLU/pool gate within float32 rounding of zero (tests/test_gpu_train.py)?"""
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
n = int(sys.argv[1])
for attempt in [int(a) for a in sys.argv[2:]]:
    h.set_params(P.params_to_flat(W))
    ax, co, sa, at = synth.patch_batch(n, seed=100 + n + 1000 * attempt)
    rng = np.random.RandomState(n + 1000 * attempt)
    y = rng.randint(0, 15, n).astype(np.uint8)
    m = {'%s_l1drop' % t: (rng.rand(n, 60, 3, 3) < 0.5).astype(np.uint8) for t in P.TOWERS}
    m['f1_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8); m['f2_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    loss_ref, gref, bn_ref, _ = O.train_forward_backward(W, ax, co, sa, at, y, m)
    m5 = np.stack([m['%s_l1drop' % t].reshape(n, 540) for t in P.TOWERS])
    pm = np.concatenate([m5.ravel(), m['f1_drop'].ravel(), m['f2_drop'].ravel()]).astype(np.uint8)
    h.train_step(dev(ax), dev(co), dev(sa), dev(at), dev(y, torch.uint8), n, masks=dev(pm, torch.uint8), stream=st)
    torch.cuda.synchronize()
    ptr, cnt = h.grad_arena()
    got = P.flat_to_params(dist.wrap_device_floats(ptr, cnt, 0).cpu().numpy().copy())
    worst, who = 0.0, ''
    for name, shapes in P.LAYER_SPEC:
        for i, _ in enumerate(shapes):
            ref = np.asarray(gref[name][i]); scale = max(np.abs(ref).max(), 1e-6)
            err = np.abs(got[name][i] - ref).max() / scale
            if err > worst: worst, who = err, name
    print('n %d attempt %d worst %.3g (%s)' % (n, attempt, worst, who), flush=True)
