"""Two independent runs of N training steps (batch 256, generated dropout masks, Adam) must end with bit-identical parameters:
a stale read in any cross-workgroup hand-off (the in-kernel batch-norm finalize, the side stream joins) would show up here as a
difference.  Also prints the loss of the last step of each run."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd')):
    sys.path.insert(0, p)
from cnn_cort import _hip, params as P, synth
W = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
SPLIT = len(sys.argv) > 3 and sys.argv[3] == 'split'        # third argument "split": options train_split (the steps) and fwd_split (a forward pass per run)
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0', dtype=dt)
ax, co, sa, at = [dev(a) for a in synth.patch_batch(B, seed=0)]
y = dev(np.random.RandomState(0).randint(0, 15, B).astype(np.uint8), torch.uint8)
st = torch.cuda.current_stream().cuda_stream
# a busy neighbour on another stream makes the load uneven (hand-offs are to be tested under uneven load)
side = torch.cuda.Stream()
big = torch.randn((4096, 4096), device='cuda:0')
out = []
for run in range(2):
    h = _hip.Handle(0); h.set_params(P.params_to_flat(W))
    if SPLIT:
        h.set_option('train_split', 1)
    for i in range(steps):
        if run == 1 and i % 3 == 0:
            with torch.cuda.stream(side):
                big = big @ big.t() * 1e-4
        loss = h.train_step(ax, co, sa, at, y, B, seed=i, want_loss=(i == steps - 1), stream=st)
        h.adam_apply(stream=st)
    torch.cuda.synchronize()
    fwd = None
    if SPLIT:                               # the trained weights through the split inference kernels, 40 000 patches, three times
        h.set_option('fwd_split', 1)
        n = 40000
        fx = [dev(a) for a in synth.patch_batch(n, seed=5)]
        probs = torch.empty((n, 15), device='cuda:0'); labels = torch.empty((n,), dtype=torch.int32, device='cuda:0')
        reps = []
        for r in range(3):
            if run == 1:
                with torch.cuda.stream(side):
                    big = big @ big.t() * 1e-4
            h.forward(fx[0], fx[1], fx[2], fx[3], n, probs, labels, st)
            torch.cuda.synchronize()
            reps.append(probs.cpu().numpy().copy())
        assert all(np.array_equal(reps[0], r) for r in reps[1:]), 'split forward differs between repetitions'
        fwd = reps[0]
    out.append((loss, h.get_params(), fwd))
    h.close()
same = np.array_equal(out[0][1], out[1][1]) and (not SPLIT or np.array_equal(out[0][2], out[1][2]))
print(('split forms, ' if SPLIT else '') + 'steps %d batch %d: last losses %r %r; parameters bit-identical: %s' % (steps, B, out[0][0], out[1][0], same))
sys.exit(0 if same and np.isfinite(out[0][1]).all() else 1)
