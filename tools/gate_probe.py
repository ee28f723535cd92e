"""Per-array gradient error of one scs_train_step against the gate-aware float64 oracle (tests/test_gpu_train.py machinery),
with the oracle's own decisions and with the GPU's:  python tools/gate_probe.py [n] [draw]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from cnn_cort import _hip, params as P, synth  # noqa: E402
from oracle import np_oracle as O  # noqa: E402
import test_gpu_train as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 37
draw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
weights = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
h = _hip.Handle(0)
h.set_params(P.params_to_flat(weights))
ax, co, sa, at = synth.patch_batch(n, seed=100 + n + 1000 * draw)
rng = np.random.RandomState(n + 1000 * draw)
y = rng.randint(0, 15, n).astype(np.uint8)
masks = T._random_masks(rng, n)
st = torch.cuda.current_stream().cuda_stream
loss = h.train_step(T.dev(ax), T.dev(co), T.dev(sa), T.dev(at), T.dev(y, torch.uint8), n, masks=T.dev(T.pack_masks(masks), torch.uint8), stream=st)
got = P.flat_to_params(T.arena(h, 'grad'))
gates = T.gpu_gates(h, n, weights, (ax, co, sa))
for label, g in (('oracle decisions', None), ('GPU decisions', gates)):
    dbg = {}
    loss_ref, grads_ref, bn_ref, _ = O.train_forward_backward(weights, ax, co, sa, at, y, masks, debug=dbg, gates=g)
    print('==', label, 'loss', loss, loss_ref, 'forced', {k: v for k, v in dbg['forced'].items() if v[0]})
    rows = []
    for name, shapes in P.LAYER_SPEC:
        for i, _ in enumerate(shapes):
            ref = np.asarray(grads_ref[name][i])
            scale = max(np.abs(ref).max(), 1e-6)
            rows.append((np.abs(got[name][i] - ref).max() / scale, name, i, scale))
    for r in sorted(rows, reverse=True)[:12]:
        print('   %.3e  %s[%d] scale %.3g' % r)
