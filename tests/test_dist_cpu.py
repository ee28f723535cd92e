"""Multi-process path on CPU (gloo, world_size 2): sharding without collectives for inference, one gradient
all-reduce per training step, lock-step schedule at a ragged epoch tail.  The compute stand-in on CPU is the
oracle (tests may use it); on the GPU the same cnn_cort.dist calls wrap the libscs gradient arena (RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG, WEIGHTS


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from cnn_cort import dist, params as P, synth
    from oracle import np_oracle as O
    r, w, _ = dist.init_from_env(backend='gloo')
    out = {'rank': r, 'world': w}
    # --- inference sharding: contiguous ranges cover the units exactly once, no collective involved
    lo, hi = dist.shard_range(1001)
    out['range'] = (lo, hi)
    out['subjects'] = dist.shard_list(['s%d' % i for i in range(5)])
    labels = np.arange(lo, hi, dtype=np.int64) * 3
    out['gathered_ok'] = bool(np.array_equal(dist.gather_concat(labels), np.arange(1001) * 3))
    # --- training schedule: 5 batches over 2 ranks -> 3 lock-step steps, the last one has a single active rank
    starts = list(range(0, 5 * 8, 8))
    sched = list(dist.shard_steps(starts))
    out['sched'] = sched
    out['active'] = [dist.active_ranks(len(starts), s) for s in range(len(sched))]
    # --- one data-parallel step: each rank back-propagates its own 4 samples, gradients are summed and scaled
    weights = P.load_params(WEIGHTS)
    ax, co, sa, at = synth.patch_batch(8, seed=9)
    y = np.arange(8) % 15
    rng = np.random.RandomState(3)
    masks = {k: (rng.rand(8, 60, 3, 3) < 0.5).astype(np.uint8) for k in ('axial_l1drop', 'coronal_l1drop', 'saggital_l1drop')}
    masks['f1_drop'] = (rng.rand(8, 540) < 0.5).astype(np.uint8)
    masks['f2_drop'] = (rng.rand(8, 540) < 0.5).astype(np.uint8)
    sl = slice(4 * r, 4 * r + 4)
    loss, grads, _, _ = O.train_forward_backward(weights, ax[sl], co[sl], sa[sl], at[sl], y[sl], {k: v[sl] for k, v in masks.items()})
    flat = torch.from_numpy(np.concatenate([np.asarray(g, np.float64).ravel() for n, _ in P.LAYER_SPEC for g in grads[n]]))
    local = flat.clone()
    two = local.clone()
    scale = dist.allreduce_flat(flat, n_active=2)
    out['avg'] = (flat * scale).numpy()
    # the two-bucket exchange (dense head's tail of the arena first, then the towers): the same sums element for element
    assert dist.allreduce_flat(two, n_active=2, first=P.head_first_float()) == scale
    out['two_buckets_equal'] = bool(torch.equal(two, flat))
    out['head_first'] = P.head_first_float()
    out['min'] = dist.min_scalar(float(r + 1))
    out['local'] = local.numpy()
    out['mean_loss'] = dist.mean_scalar(loss, 1)
    out['loss'] = loss
    out['max'] = dist.max_scalar(float(r + 1))
    dist.barrier()
    q.put(out)


@pytest.mark.timeout(600)
def test_two_rank_gloo_data_parallel_plumbing():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda d: d['rank'])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res
    assert (a['world'], b['world']) == (2, 2)
    assert a['range'] == (0, 501) and b['range'] == (501, 1001)
    assert a['subjects'] == ['s0', 's2', 's4'] and b['subjects'] == ['s1', 's3']
    assert a['gathered_ok'] and b['gathered_ok']
    assert a['sched'] == [(0, True), (16, True), (32, True)] and b['sched'] == [(8, True), (24, True), (None, False)]
    assert a['active'] == [2, 2, 1]
    assert np.array_equal(a['avg'], b['avg'])                                    # replicas see identical gradients
    assert np.allclose(a['avg'], 0.5 * (a['local'] + b['local']), rtol=0, atol=1e-15)
    assert abs(a['mean_loss'] - 0.5 * (a['loss'] + b['loss'])) < 1e-12 and a['max'] == 2.0
    assert a['two_buckets_equal'] and b['two_buckets_equal'] and a['head_first'] == 436320 and (a['min'], b['min']) == (1.0, 1.0)


class _HostHandle:
    """Stand-in for a libscs handle in the gloo rehearsal: host gradient / parameter arenas with the real layout constants."""

    def __init__(self, n, first):
        self.g = torch.zeros(n, dtype=torch.float32)
        self.p = torch.zeros(n, dtype=torch.float32)
        self.first = first
        self.waited = 0

    def grad_arena_tensor(self):
        return self.g

    def param_arena_tensor(self):
        return self.p

    def wait_head_grads(self, stream):          # (_hip.Handle.wait_head_grads: first float and count of the dense head's bucket)
        self.waited += 1
        return self.first, self.g.numel() - self.first


def _tail_worker(rank, world, port, q, buckets):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      SCS_DP_BUCKETS=str(buckets))
    from cnn_cort import dist, params as P
    dist.init_from_env(backend='gloo')
    n, first = P.N_STORED, P.head_first_float()
    h = _HostHandle(n, first)
    ramp = torch.arange(n, dtype=torch.float32) * 1e-6
    starts = [0, 8, 16]                              # three mini-batches over two ranks: step 1 has ONE active rank (rank 0)
    got = []
    for step, (start, active) in enumerate(dist.shard_steps(starts)):
        def fill():
            if active:                               # "this step's gradients": a ramp that identifies rank and step in every element
                h.g.copy_(ramp + (10.0 * (step + 1) + rank))
            else:
                h.g.fill_(123.0)                     # stale gradients of the previous step: must not reach the sum
        # once with the schedule known (what Net.fit passes), once asking the ranks (the extra flag exchange)
        fill()
        scale_known = dist.allreduce_grads(h, active, 0, n_active=dist.active_ranks(len(starts), step, world))
        summed = h.g.clone().numpy()
        fill()
        scale_asked = dist.allreduce_grads(h, active, 0)
        got.append((bool(active), scale_known, scale_asked, summed, h.g.clone().numpy()))
    h.p.fill_(float(rank + 1))                       # the epoch's last collective, behind the ragged tail: the arena average
    dist.sync_params(h)
    dist.barrier()
    q.put({'rank': rank, 'steps': got, 'waited': h.waited, 'buckets': dist.dp_buckets(), 'p': (float(h.p.min()), float(h.p.max()))})


@pytest.mark.timeout(600)
@pytest.mark.parametrize('buckets', [2, 1])
def test_ragged_epoch_tail_in_the_bucketed_exchange(buckets):
    """SURVEY 8(e) at the tail of an epoch, both bucket forms, REAL collectives in their real order (cnn_cort/dist.py,
    allreduce_grads): three mini-batches over two ranks leave the second step with one active rank.  The inactive rank must zero its
    stale arena and issue the same collectives in the same order as the active one - dense head's bucket, towers' bucket, (flag) -
    or gloo pairs a 447 135-float all-reduce with a 436 320-float one and the exchange fails; the sums must be the active ranks'
    gradients element for element and the scale 1 / active ranks, whether the schedule is passed in or asked for; the parameter
    average of the epoch's end follows on the same group."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_tail_worker, args=(r, world, port, q, buckets)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda d: d['rank'])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res
    assert a['buckets'] == b['buckets'] == buckets
    from cnn_cort import params as P
    ramp = (torch.arange(P.N_STORED, dtype=torch.float32) * 1e-6)
    # step 0: both active -> sum of the two ramps, scale 1/2; step 1: rank 0 alone -> its ramp, scale 1
    want0 = ((ramp + 10.0) + (ramp + 11.0)).numpy()
    want1 = (ramp + 20.0).numpy()
    for r in (a, b):
        (act0, k0, s0, sum0, again0), (act1, k1, s1, sum1, again1) = r['steps']
        assert act0 and act1 == (r['rank'] == 0)
        assert (k0, s0, k1, s1) == (0.5, 0.5, 1.0, 1.0)
        assert np.array_equal(sum0, want0) and np.array_equal(again0, want0)
        assert np.array_equal(sum1, want1) and np.array_equal(again1, want1)          # the stale 123s of the idle rank are gone
        assert r['p'] == (1.5, 1.5)
    # an active rank asks libscs where the dense head's bucket starts (scs_wait_head_grads) exactly when it had a step of its own
    assert (a['waited'], b['waited']) == ((4, 2) if buckets == 2 else (0, 0))


def test_single_process_degrades_to_noops():
    from cnn_cort import dist
    assert dist.world_size() == 1 and dist.rank() == 0
    assert dist.shard_range(10) == (0, 10) and dist.shard_list([1, 2, 3]) == [1, 2, 3]
    assert list(dist.shard_steps([0, 4, 8])) == [(0, True), (4, True), (8, True)]
    t = torch.ones(4)
    assert dist.allreduce_flat(t, 1) == 1.0 and dist.mean_scalar(3.0, 2) == 3.0
    dist.barrier()


def test_bucket_switch_and_head_bucket_layout(monkeypatch):
    """SCS_DP_BUCKETS is read at every exchange (1 unless it says 2), and the dense head's bucket a rank WITHOUT a step of its own
    exchanges (dist.handle_head_bucket) is the arena's tail from FC1.W on - what scs_wait_head_grads reports to the active ranks."""
    from cnn_cort import dist, params as P
    monkeypatch.delenv('SCS_DP_BUCKETS', raising=False)
    assert dist.dp_buckets() == 1
    for val, want in (('2', 2), ('1', 1), ('3', 1), ('x', 1)):
        monkeypatch.setenv('SCS_DP_BUCKETS', val)
        assert dist.dp_buckets() == want
    first, count = dist.handle_head_bucket(P.N_STORED)
    assert first == P.flat_offsets()[('FC1', 0)][0] == 436320 and first + count == P.N_STORED
    names = [n for n, _ in P.LAYER_SPEC]
    assert names.index('FC1') > max(names.index('%s_prelu_d1' % t) for t in P.TOWERS)       # everything of the towers lies in front of it
