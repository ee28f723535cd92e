"""
Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) on xGMI.

The reference is single-process / single-device ("Parallelism strategies: none"), so this
layer is new.  What shards, and how:
  * inference: patches / subjects are independent -> contiguous slices of the centre list (shard_range) or
    whole subjects (shard_list) per rank, NO collective on the data path.
  * training: pure data parallelism, identical replicas; per step each rank runs one mini-batch and ONE
    exchange happens - a sum all-reduce of the contiguous float32 gradient arena (883 455 floats, 3.5 MB),
    scaled by 1/active_ranks inside the Adam kernel.  Ranks without a batch at the tail of an epoch
    contribute zeros.  BN running statistics are local per step and reconciled by an arena average once per
    epoch (sync_params).
All functions degrade to no-ops when torch.distributed is not initialised (world size 1).
"""
import os

import numpy as np


def _td():
    import torch.distributed as td
    return td


def is_dist():
    try:
        td = _td()
        return td.is_available() and td.is_initialized()
    except Exception:
        return False


def world_size():
    return _td().get_world_size() if is_dist() else 1


def rank():
    return _td().get_rank() if is_dist() else 0


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE, MASTER_*).
    backend defaults to 'nccl' (= RCCL) when a GPU is visible, else 'gloo'.  Returns (rank, world, local_rank)."""
    import torch
    td = _td()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or os.environ.get('SCS_DIST_FORCE_INIT')) and not td.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('SCS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            # one GPU per rank; LOCAL_RANK wraps around only in single-GPU rehearsals (gloo backend)
            # (ranks of THIS node: a 2-node x 8-GPU job has WORLD_SIZE 16 and LOCAL_WORLD_SIZE 8)
            local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
            if backend == 'nccl' and (local_world > torch.cuda.device_count() or local >= torch.cuda.device_count()):
                raise RuntimeError('backend nccl (RCCL) needs one GPU per rank: %d ranks on this node, %d visible device(s)'
                                   % (local_world, torch.cuda.device_count()))
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
        td.init_process_group(backend=backend, rank=int(os.environ.get('RANK', '0')), world_size=world)
    if torch.cuda.is_available():
        local = local % max(1, torch.cuda.device_count())
    return rank(), world_size(), local


# ----------------------------------------------------------------------------------- sharding (no comms)
def shard_range(n, world=None, r=None):
    """Contiguous [lo, hi) slice of n independent units for rank r (sizes differ by at most 1)."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    base, rem = divmod(int(n), world)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def shard_list(items, world=None, r=None):
    """Round-robin assignment of whole items (subjects) to ranks: items[r::world]."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    return list(items)[r::world]


def shard_steps(starts, world=None, r=None):
    """Lock-step schedule of mini-batches: step s gives rank r the batch starts[s*world + r].
    Yields (start_or_None, active) for every step so that all ranks run the same number of collectives."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    nb = len(starts)
    for s in range((nb + world - 1) // world):
        i = s * world + r
        yield (starts[i], True) if i < nb else (None, False)


def active_ranks(n_batches, step, world=None):
    world = world_size() if world is None else world
    return max(0, min(world, n_batches - step * world))


# ----------------------------------------------------------------------------------- collectives
class _DevArray:
    """Zero-copy view of a raw device pointer for torch (CUDA array interface v2)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


_wrapped = {}


def wrap_device_floats(ptr, n, device):
    """torch view of `n` floats at device pointer `ptr` (cached: the arenas live as long as their handle)."""
    import torch
    key = (int(ptr), int(n), int(device))
    t = _wrapped.get(key)
    if t is None:
        t = torch.as_tensor(_DevArray(ptr, n), device='cuda:%d' % device)
        if len(_wrapped) > 64:
            _wrapped.clear()
        _wrapped[key] = t
    return t


def allreduce_flat(t, n_active, first=None):
    """Sum all-reduce of a flat tensor (any backend); returns the scale that turns the sum into the mean over
    the ranks that contributed.  first: exchange as two buckets, t[first:] then t[:first] (the order allreduce_grads uses:
    the dense head's tail of the arena, then the towers) - element for element the same sums."""
    if is_dist() and world_size() > 1:
        td = _td()
        if first is None:
            td.all_reduce(t, op=td.ReduceOp.SUM)
        else:
            td.all_reduce(t[first:], op=td.ReduceOp.SUM)
            td.all_reduce(t[:first], op=td.ReduceOp.SUM)
    return 1.0 / max(1, n_active)


_comm_streams = {}


def dp_buckets():
    """Buckets of the gradient exchange: 1 (default: one all-reduce of the whole arena behind the step) or 2 (SCS_DP_BUCKETS=2:
    the dense head's bucket is exchanged on a communication stream while the conv backward pass still runs)."""
    try:
        return 2 if int(os.environ.get('SCS_DP_BUCKETS', '1')) == 2 else 1
    except ValueError:
        return 1


def allreduce_grads(handle, active, stream=0, n_active=None, buckets=None):
    """All-reduce the gradient arena of a libscs handle across ranks (RCCL); returns grad_scale for Adam.
    `active` False = this rank had no batch this step: its arena is zeroed first.

    buckets = 2 (default from SCS_DP_BUCKETS): the arena goes in two collectives, in the same order on every rank -
    first its tail, the dense head's gradients (FC1 / fc_2 / out_layer, 51 % of the floats), which scs_train_step finishes
    ~0.3 ms before the step ends: a communication stream waits for exactly that (scs_wait_head_grads) and the all-reduce is issued
    there, so it runs beside the conv backward pass; then the towers' bucket on the caller's stream behind the step.  The
    caller's stream waits for the first bucket before this returns (scs_adam_apply is enqueued behind both).  The sums are the
    same element by element: one bucket or two give the same gradients."""
    if not is_dist():                         # a process group of ONE rank still runs the collective (RCCL rehearsal on one GPU)
        return 1.0
    import torch
    g = _grad_arena_tensor(handle)
    n = g.numel()
    td = _td()
    if buckets is None:
        buckets = dp_buckets()
    if buckets == 2 and active and not g.is_cuda:
        # host arena (the gloo rehearsal of tests/test_dist_cpu.py): no streams to overlap on, the SAME collectives in the same order
        first, count = handle.wait_head_grads(0)
        assert first + count == n
        td.all_reduce(g[first:], op=td.ReduceOp.SUM)
        td.all_reduce(g[:first], op=td.ReduceOp.SUM)
    elif buckets == 2 and active:
        dev = g.device
        comm = _comm_streams.get(dev)
        if comm is None:
            comm = _comm_streams[dev] = torch.cuda.Stream(device=dev)
        first, count = handle.wait_head_grads(comm.cuda_stream)
        assert first + count == n
        with torch.cuda.stream(comm):
            td.all_reduce(g[first:], op=td.ReduceOp.SUM)           # issued on (behind) the communication stream
            done = torch.cuda.Event()
            done.record(comm)
        td.all_reduce(g[:first], op=td.ReduceOp.SUM)               # the towers: behind the step on the caller's stream
        torch.cuda.current_stream(dev).wait_event(done)
    else:
        if not active:
            g.zero_()
        if buckets == 2:                                           # same collectives, same order as the active ranks
            first, count = handle_head_bucket(n)
            td.all_reduce(g[first:], op=td.ReduceOp.SUM)
            td.all_reduce(g[:first], op=td.ReduceOp.SUM)
        else:
            td.all_reduce(g, op=td.ReduceOp.SUM)
    if n_active is None:                      # callers that know the schedule pass it and skip this exchange
        flag = torch.tensor([1.0 if active else 0.0], device=g.device)
        td.all_reduce(flag, op=td.ReduceOp.SUM)
        n_active = int(flag.item())
    return 1.0 / max(1, n_active)


def _grad_arena_tensor(handle):
    """The gradient arena of a handle as a flat torch tensor: a zero-copy view of libscs's device arena, or - for a stand-in handle
    in the CPU rehearsals - whatever its grad_arena_tensor() returns."""
    t = getattr(handle, 'grad_arena_tensor', None)
    if t is not None:
        return t()
    ptr, n = handle.grad_arena()
    return wrap_device_floats(ptr, n, handle.device)


def handle_head_bucket(n):
    """(first float, count) of the dense head's bucket in an arena of n floats - the layout constant a rank WITHOUT a step of
    its own needs (an active rank gets it from scs_wait_head_grads)."""
    from . import params as P
    first = P.head_first_float()
    return first, n - first


def sync_params(handle):
    """Average the parameter arena across ranks (reconciles the per-rank BN running statistics; the trainable
    entries are already identical)."""
    if not is_dist():
        return
    t = getattr(handle, 'param_arena_tensor', None)            # (stand-in handles of the CPU rehearsals, as _grad_arena_tensor)
    if t is not None:
        p = t()
    else:
        ptr, n = handle.param_arena()
        p = wrap_device_floats(ptr, n, handle.device)
    td = _td()
    td.all_reduce(p, op=td.ReduceOp.SUM)
    p.mul_(1.0 / world_size())


def mean_scalar(value, count):
    """Count-weighted mean of a per-rank scalar (epoch train loss)."""
    if world_size() == 1:
        return float(value)
    import torch
    td = _td()
    dev = 'cuda' if td.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value) * count, float(count)], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t[0] / t[1]) if float(t[1]) > 0 else 0.0


def max_scalar(value):
    """Max over ranks (bench timing)."""
    if world_size() == 1:
        return float(value)
    import torch
    td = _td()
    dev = 'cuda' if td.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t[0])


def min_scalar(value):
    """Min over ranks (bench: the fastest rank's time next to the slowest's, so stragglers show)."""
    if world_size() == 1:
        return float(value)
    import torch
    td = _td()
    dev = 'cuda' if td.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.MIN)
    return float(t[0])


def sum_scalar(value):
    if world_size() == 1:
        return float(value)
    import torch
    td = _td()
    dev = 'cuda' if td.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t[0])


def sum_pair(a, b):
    """Sum of two per-rank scalars over the ranks in ONE exchange (epoch loss sums and their counts)."""
    if world_size() == 1:
        return float(a), float(b)
    import torch
    td = _td()
    dev = 'cuda' if td.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(a), float(b)], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t[0]), float(t[1])


def barrier():
    if is_dist() and world_size() > 1:
        _td().barrier()


def gather_concat(arr, world=None):
    """All-gather variable-length numpy arrays (labels of a sharded centre list) and concatenate in rank order.
    Only needed when ONE rank must own the whole output; the compute path itself has no collective."""
    if world_size() == 1:
        return arr
    td = _td()
    out = [None] * world_size()
    td.all_gather_object(out, np.asarray(arr))
    return np.concatenate(out, axis=0)
