"""How long does the HOST take to issue one training step (scs_train_step + scs_adam_apply, no synchronisation)?  If this is
close to the measured step time, the step is bound by launch issue, not by the GPU."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd')):
    sys.path.insert(0, p)
from cnn_cort import _hip, params as P, synth
W = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
h = _hip.Handle(0); h.set_params(P.params_to_flat(W))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0', dtype=dt)
ax, co, sa, at = [dev(a) for a in synth.patch_batch(B, seed=0)]
y = dev(np.random.RandomState(0).randint(0, 15, B).astype(np.uint8), torch.uint8)
st = torch.cuda.current_stream().cuda_stream
FUSED = os.environ.get('SCS_ISSUE_FUSED', '1') != '0'          # the update inside the step (what bench.py times on one rank)
if FUSED:
    h.set_fused_adam(True)
def one(i):
    h.train_step(ax, co, sa, at, y, B, seed=i, want_loss=False, stream=st)
    if not FUSED:
        h.adam_apply(stream=st)
for _ in range(20):
    one(1)
torch.cuda.synchronize()
for steps in (50, 200):
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('batch %d, %d steps: host issue %.3f ms/step, until the GPU is done %.3f ms/step' % (B, steps, 1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps))
