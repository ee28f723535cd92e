"""Worker of tests/test_gpu_rccl.py: the first RCCL calls this code ever makes, on one GPU.

Runs three training steps (scs_train_step + gradient exchange + scs_adam_apply + parameter reconciliation) twice on the same
inputs - once without any process group, once inside an RCCL (backend 'nccl') process group of ONE rank - and checks that the
parameters, Adam-updated arenas and BN running statistics agree bit for bit: a one-rank sum all-reduce is the identity, so any
difference is a plumbing fault (communicator creation, the __cuda_array_interface__ view of the arenas, the ordering of
RCCL's stream against the library's launches on torch's current stream).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd')):
    sys.path.insert(0, p)
from cnn_cort import _hip, dist, params as P  # noqa: E402


def run_steps(flat, n=64, steps=3):
    g = torch.Generator(device='cuda:0')
    views = []
    for i in range(3):
        g.manual_seed(i)
        views.append(torch.randn((n, 1, 32, 32), generator=g, device='cuda:0'))
    g.manual_seed(3)
    atlas = torch.rand((n, 15), generator=g, device='cuda:0')
    atlas = (atlas / atlas.sum(1, keepdim=True)).contiguous()
    y = (torch.arange(n, device='cuda:0') % 15).to(torch.uint8)
    h = _hip.Handle(0)
    h.set_params(flat)
    st = torch.cuda.current_stream().cuda_stream
    grads = []
    for s in range(steps):
        h.train_step(views[0], views[1], views[2], atlas, y, n, seed=77 + s, want_loss=False, stream=st)
        scale = dist.allreduce_grads(h, True, st, n_active=1)          # RCCL all-reduce of the 3.5 MB arena when a group exists
        assert scale == 1.0
        ptr, cnt = h.grad_arena()
        grads.append(dist.wrap_device_floats(ptr, cnt, 0).clone())     # ordered on the current stream behind the collective
        h.adam_apply(grad_scale=scale, stream=st)
    dist.sync_params(h)                                                # arena average over the (one) rank
    torch.cuda.synchronize()
    out = h.get_params(), [t.cpu().numpy() for t in grads]
    h.close()
    return out


def main():
    flat = P.params_to_flat(P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1')))
    assert not dist.is_dist()
    p0, g0 = run_steps(flat)
    os.environ.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', SCS_DIST_FORCE_INIT='1')
    os.environ.setdefault('MASTER_PORT', '29631')
    rank, world, local = dist.init_from_env(backend='nccl')
    assert dist.is_dist() and torch.distributed.get_backend() == 'nccl' and (rank, world, local) == (0, 1, 0)
    assert int(round(dist.sum_scalar(1.0))) == 1 and dist.max_scalar(2.5) == 2.5
    p1, g1 = run_steps(flat)
    assert all(np.array_equal(a, b) for a, b in zip(g0, g1)), 'gradient arena differs behind a one-rank RCCL all-reduce'
    assert np.array_equal(p0, p1), 'parameters differ: %g' % float(np.abs(p0 - p1).max())
    # the two-bucket exchange (SCS_DP_BUCKETS=2): the dense head's bucket is all-reduced on a communication stream that waits
    # for scs_wait_head_grads' event - issued right behind scs_train_step, while its kernels are still running - the towers'
    # bucket behind the step; scs_adam_apply is ordered behind both.  Same bits.
    os.environ['SCS_DP_BUCKETS'] = '2'
    assert dist.dp_buckets() == 2
    p2, g2 = run_steps(flat, steps=6)
    del os.environ['SCS_DP_BUCKETS']
    assert all(np.array_equal(a, b) for a, b in zip(g0, g2[:3])), 'gradient arena differs with the two-bucket exchange'
    p3, g3 = run_steps(flat, steps=6)
    assert all(np.array_equal(a, b) for a, b in zip(g3, g2)) and np.array_equal(p3, p2), 'two buckets: parameters differ after 6 steps'
    assert not np.array_equal(p0, flat)                                # the steps did move the parameters
    # a bare collective on a torch tensor, for the record of what the communicator does
    t = torch.arange(8, dtype=torch.float32, device='cuda:0')
    torch.distributed.all_reduce(t)
    torch.cuda.synchronize()
    assert t.tolist() == list(range(8))
    dist.barrier()
    torch.distributed.destroy_process_group()
    print('RCCL_ONE_RANK_OK nccl world 1: parameters and gradients bit-identical to the run without a process group')


if __name__ == '__main__':
    main()
