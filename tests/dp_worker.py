"""Worker of tests/test_gpu_train.py::test_two_rank_gradient_exchange (launched with torch.distributed.run)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd')):
    sys.path.insert(0, p)
from cnn_cort import _hip, dist, params as P  # noqa: E402


def main(out_dir):
    rank, world, local = dist.init_from_env()
    torch.cuda.set_device(local)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g5_train.npz'))
    n = 16
    sl = slice(rank * n, rank * n + n)
    masks = np.concatenate([np.stack([g['mask_%s_l1drop' % t][sl].reshape(n, 540) for t in P.TOWERS]).ravel(),
                            g['mask_f1_drop'][sl].ravel(), g['mask_f2_drop'][sl].ravel()]).astype(np.uint8)
    dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:%d' % local, dtype=dt)
    h = _hip.Handle(local)
    h.set_params(P.params_to_flat(P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))))
    st = torch.cuda.current_stream().cuda_stream
    h.train_step(dev(g['ax'][sl]), dev(g['co'][sl]), dev(g['sa'][sl]), dev(g['atlas'][sl]), dev(g['y'][sl], torch.uint8), n,
                 masks=dev(masks, torch.uint8), stream=st)
    ptr, cnt = h.grad_arena()
    torch.cuda.synchronize()
    local_grad = dist.wrap_device_floats(ptr, cnt, local).cpu().numpy().copy()
    scale = dist.allreduce_grads(h, True, st, n_active=world)
    torch.cuda.synchronize()
    summed = dist.wrap_device_floats(ptr, cnt, local).cpu().numpy().copy()
    h.adam_apply(grad_scale=scale, stream=st)
    dist.sync_params(h)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), local=local_grad, summed=summed, scale=scale, params=h.get_params())
    dist.barrier()
    h.close()


if __name__ == '__main__':
    main(sys.argv[1])
