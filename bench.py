#!/usr/bin/env python3
"""
bench.py - headline benchmark of the hot path (BASELINE.json): tri-view 32x32 patches/s.

  python bench.py --gpus N --steps K --warmup W [--mode forward|train] [--batch B]

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
  forward (default; BASELINE configs[1]): deterministic forward of B = 100 000 tri-view patches through
      scs_forward (probabilities + labels), shipped miccai2012_v1 weights.
  train   (BASELINE configs[2]/[3]):     one scs_train_step + gradient all-reduce (RCCL, N > 1) + scs_adam_apply
      on B = 256 samples per GPU.
N > 1: one rank per GPU over RCCL.  Either launched by torchrun (RANK/WORLD_SIZE/... in the environment) or, when
`--gpus N` is given without that environment, bench.py starts the N ranks itself as child processes (torchrun,
127.0.0.1) before touching the GPU; a world size that differs from --gpus is an error, never a silent 1-GPU run.
Ranks process independent batches (weak scaling), the forward
path has no collective, the training path exactly one (the gradient arena).  Timing = K steps between
barrier + synchronize brackets, max over ranks; rank 0 prints ONE JSON line.

The default (forward) run also measures the training step once its own timed region is over - `train_step`: batch 256
(BASELINE configs[2]) and 128 (the reference's effective batch), 200 steps each of scs_train_step + all-reduce +
scs_adam_apply, with the dominant training kernel's roofline - so that the driver's one line carries both halves of
BASELINE.json's metric ("fwd & train step").  The headline keys stay the forward measurement.

roofline: the dominant kernel (forward: fused conv1+conv2+pool1, 50.6 % of the net's MACs; train: the longest of the
conv forward / data-gradient / weight-gradient kernels) is timed live with HIP events on the stream it is launched on
(scs_forward_profiled / scs_train_step_profiled); achieved = algorithmic FLOP per launch / mean launch duration against
the 157.3 TFLOP/s fp32 matrix/vector peak of gfx950.  cpu_baseline: the float32 torch-CPU port of the same net
(oracle/torch_ref.py, all host threads) on a bounded sample - a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'sub-cortical_segmentation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# MACs per tri-view patch and stage (nets.py:170-231), all three towers
STAGE_NAMES = ['conv1+conv2+pool1', 'conv3', 'conv4+pool2', 'conv5', 'd1x3', 'FC1', 'fc_2', 'out+softmax']
STAGE_MACS = [3 * (162000 + 2822400), 3 * 1036800, 3 * 1440000, 3 * 194400, 3 * 97200, 291600, 149850, 4050]
FWD_FLOP = 2 * sum(STAGE_MACS)                 # 35 407 800
TRAIN_FLOP = 3 * FWD_FLOP                      # fwd + dgrad + wgrad
PEAK_F32_TFLOPS = 157.3                        # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_F16_TFLOPS = 2500.0                       # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak
TRAIN_SPLIT_DTYPE = ('f32; the forward pass of the conv1 + conv2 block with f32 operands as two f16 pieces each, three f16 MFMA products per '
                     'multiply-add block, f32 accumulate (option train_split); every other kernel on the exact-f32 MFMA')
SPLIT_DTYPE = ('f32 operands as two f16 pieces each, three f16 MFMA products per multiply-add block, f32 accumulate (option fwd_split: '
               'conv2..conv5 and the dense head d1 / FC1 / fc_2; conv1 and the 270 x 15 out layer in exact f32)')
ALGO_BYTES_FWD = 3 * 4096 + 60 + 60            # patches + atlas in, probs out


def synth_on_device(torch, n, seed, dev):
    g = torch.Generator(device=dev)
    views = []
    for i in range(3):
        g.manual_seed(seed + i)
        views.append(torch.randn((n, 1, 32, 32), generator=g, device=dev, dtype=torch.float32))
    g.manual_seed(seed + 3)
    e = -torch.log(torch.rand((n, 15), generator=g, device=dev, dtype=torch.float32).clamp_min(1e-12))
    atlas = (e / e.sum(1, keepdim=True)).contiguous()                 # Dirichlet(1_15)
    return views[0], views[1], views[2], atlas


def collective_config(world):
    """What a SCALE reader needs beside the numbers: the bucket form of the gradient exchange (SCS_DP_BUCKETS: 1 = one all-reduce of
    the arena behind the step, 2 = the dense head's bucket on a communication stream under the conv backward pass), the collective
    library's version and the transport switches that were set."""
    from cnn_cort import dist
    out = {'dp_buckets': dist.dp_buckets(), 'world': world}
    try:
        import torch
        v = torch.cuda.nccl.version()                                  # RCCL's version on ROCm builds
        out['rccl_version'] = '.'.join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        out['torch'] = torch.__version__
    except Exception as e:
        out['rccl_version'] = 'unavailable (%s)' % type(e).__name__
    env = {k: os.environ[k] for k in ('SCS_DP_BUCKETS', 'NCCL_ALGO', 'NCCL_PROTO', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'RCCL_MSCCL_ENABLE',
                                      'HSA_ENABLE_IPC_MODE_LEGACY', 'NCCL_P2P_DISABLE', 'NCCL_DEBUG') if k in os.environ}
    if env:
        out['env'] = env
    return out


def usable_cpus():
    """Host threads this process may actually run on: affinity mask capped by the cgroup CPU quota (the GPU boxes
    expose 256 logical CPUs but grant a 16-CPU share; 256 torch threads on that share run 100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


CPU_BUDGET_S = float(os.environ.get('SCS_BENCH_CPU_BUDGET', '12'))     # seconds of host work per cpu_baseline sample


def cpu_baseline(weights, budget_s=None, batch=128):
    """float32 torch-CPU port on the host cores, bounded sample (kind 'port': the reference's Theano CPU path
    cannot run here - BASELINE.md section 3)."""
    import torch
    from oracle import torch_ref as T
    from cnn_cort import synth
    budget_s = CPU_BUDGET_S if budget_s is None else budget_s
    cores = usable_cpus()
    torch.set_num_threads(cores)
    net = T.FoldedCPUNet(weights)
    ax, co, sa, at = [torch.from_numpy(a) for a in synth.patch_batch(batch, seed=0)]
    net.forward(ax, co, sa, at)                                        # warm-up
    done, t0 = 0, time.time()
    while time.time() - t0 < budget_s:
        net.forward(ax, co, sa, at)
        done += batch
    dt = time.time() - t0
    return {'value': done / dt, 'unit': 'patches/s', 'cores': cores, 'kind': 'port',
            'sample': '%d N(0,1) tri-view patches in batches of %d (nolearn effective batch), float32 torch-CPU forward, %.1f s'
                      % (done, batch, dt)}


def cpu_baseline_volume(weights, budget_s=None, chunk=4096, batch=128):
    """Whole-subject stand-in: the literal per-voxel patch extraction of
    load_patch_batch / get_patches (oracle/patches_ref.py; one np.pad of the volume per view and chunk, then one slice
    per voxel - the reference does this per 100 000-voxel chunk) + the torch-CPU forward in nolearn's batches of 128,
    on a bounded sample of ROI voxels of a synthetic 256x256x192 volume."""
    import torch
    from oracle import patches_ref as R, torch_ref as T
    from cnn_cort import synth
    budget_s = CPU_BUDGET_S if budget_s is None else budget_s
    cores = usable_cpus()
    torch.set_num_threads(cores)
    shape = (256, 256, 192)
    vol = R.zscore_inference(synth.smooth_volume(shape, seed=0, as_int=True))
    rng = np.random.RandomState(0)
    net = T.FoldedCPUNet(weights)
    done, t0 = 0, time.time()
    while done == 0 or time.time() - t0 < budget_s:
        centers = [tuple(int(rng.randint(40, s - 40)) for s in shape) for _ in range(chunk)]
        views = [np.stack([np.array(R.get_patches(vol, centers, (32, 32), mode=m)).astype(np.float32)], axis=1)
                 for m in ('axial', 'coronal', 'saggital')]
        atlas_rows = rng.dirichlet(np.ones(15), size=chunk).astype(np.float32)
        for lo in range(0, chunk, batch):
            net.forward(*[torch.from_numpy(v[lo:lo + batch]) for v in views], torch.from_numpy(atlas_rows[lo:lo + batch]))
        done += chunk
    dt = time.time() - t0
    return {'value': done / dt, 'unit': 'patches/s', 'cores': cores, 'kind': 'port',
            'sample': '%d ROI voxels of a 256x256x192 volume in chunks of %d: literal get_patches per view + float32 torch-CPU '
                      'forward in batches of %d, %.1f s (a 400 000-voxel subject would take %.0f s)'
                      % (done, chunk, batch, dt, 400000 * dt / done)}


def cpu_baseline_train(weights, budget_s=None, batch=256):
    """Training-step stand-in on the host cores: the torch-CPU restatement (autograd backward) + Adam, float32,
    batch 256, explicit dropout masks - BASELINE.md section 3, config 3."""
    import torch
    from oracle import np_oracle as O, torch_ref as T
    from cnn_cort import params as P, synth
    budget_s = CPU_BUDGET_S if budget_s is None else budget_s
    cores = usable_cpus()
    torch.set_num_threads(cores)
    ax, co, sa, at = synth.patch_batch(batch, seed=0)
    rng = np.random.RandomState(4)
    y = rng.randint(0, 15, batch)
    masks = {'%s_l1drop' % t: (rng.rand(batch, 60, 3, 3) < 0.5).astype(np.float32) for t in P.TOWERS}
    masks['f1_drop'] = (rng.rand(batch, 540) < 0.5).astype(np.float32)
    masks['f2_drop'] = (rng.rand(batch, 540) < 0.5).astype(np.float32)
    params, state = weights, {}
    done, t0 = 0, time.time()
    while done == 0 or time.time() - t0 < budget_s:
        _, grads, bn, _ = T.train_loss_and_grads(params, ax, co, sa, at, y, masks, dtype=torch.float32)
        params = O.bn_running_update(O.adam_step(params, grads, state, trainable=P.is_trainable), bn)
        done += batch
    dt = time.time() - t0
    return {'value': done / dt, 'unit': 'patches/s', 'cores': cores, 'kind': 'port',
            'sample': '%d training steps of batch %d (fwd + autograd bwd + Adam), float32 torch-CPU restatement, %.1f s'
                      % (done // batch, batch, dt)}


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources: ties a committed PMC measurement to the code it was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(PKG, 'csrc', '*.h')) + glob.glob(os.path.join(PKG, 'csrc', '*.hip'))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, 'rb').read())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """HBM bytes per launch of kernel `key` from the committed rocprofv3 --pmc passes (profiles/traffic.json, written by
    tools/pmc_traffic.py) - reported only when that file was measured on the kernel sources of THIS tree; otherwise null plus
    a note saying which sources it belongs to (a constant that silently outlives the kernel it describes is worse than none)."""
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        t = json.load(open(tfile))
    except Exception:
        return None, 'profiles/traffic.json missing'
    here = csrc_sha16()
    if t.get('_csrc_sha16') != here:
        return None, 'profiles/traffic.json was measured on kernel sources %s (commit %s); this tree has %s' % (
            t.get('_csrc_sha16'), t.get('_source_commit'), here)
    return t.get(key), 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, commit %s (tools/pmc_traffic.py)' % t.get('_source_commit')


TRAIN_CONV_MACS = {'conv1+conv2 fwd': 3 * (162000 + 2822400), 'conv3 fwd': 3 * 1036800, 'conv4 fwd': 3 * 1440000, 'conv5 fwd': 3 * 194400,
                   'conv5 dgrad': 3 * 194400, 'conv4 dgrad': 3 * 1440000, 'conv3 dgrad': 3 * 1036800,
                   'conv2 dgrad + layer-1 sums': 3 * (2822400 + 162000), 'conv5 wgrad': 3 * 194400, 'conv4 wgrad': 3 * 1440000,
                   'conv3 wgrad': 3 * 1036800, 'conv2 wgrad': 3 * 2822400,
                   'conv5 backward, one launch (dgrad + wgrad + bn4 bwd sums)': 3 * 2 * 194400,
                   'conv5 wgrad partial (side stream)': 3 * 194400}
C5B_STAGE, C5W_STAGE = 'conv5 backward, one launch (dgrad + wgrad + bn4 bwd sums)', 'conv5 wgrad partial (side stream)'


def train_inputs(torch, B, rank, dev):
    ax, co, sa, at = synth_on_device(torch, B, 1000 * rank, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(4 + rank)
    y = torch.randint(0, 15, (B,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    return ax, co, sa, at, y


def train_stage_profile(h, inputs, B, st, reps=5):
    """Every launch of one training step timed with HIP events on the stream it is issued on (scs_train_step_profiled; the
    weight-gradient kernels run on the library's side stream), median of a few steps -> (roofline of the dominant conv
    kernel, per-stage table, launches per step, sum of launch times)."""
    ax, co, sa, at, y = inputs
    seen = {}
    for r in range(reps):
        for k, v in h.train_step_profiled(ax, co, sa, at, y, B, seed=1000 + r, stream=st).items():
            seen.setdefault(k, []).append(v)
    # median over the repetitions: the first launch of a step sits behind an empty queue, so a slow enqueue on the host (the
    # events are recorded by the GPU when it reaches them) shows up in its time now and then
    acc = {k: float(sorted(v)[len(v) // 2]) for k, v in seen.items()}
    macs = dict(TRAIN_CONV_MACS)
    if acc.get(C5W_STAGE, 0.0) > 0:                # the conv5 block's backward ran as two launches: the first is the data path only
        macs[C5B_STAGE] = 3 * 194400
    convs = {k: v for k, v in macs.items() if acc.get(k, 0.0) > 0}
    dom = max(convs, key=lambda k: acc[k])
    dur = acc[dom] * 1e-3
    achieved = 2.0 * convs[dom] * B / dur / 1e12
    traffic, note = measured_traffic('train: ' + dom) if B == 256 else (None, 'measured at batch 256 only')
    roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': achieved, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_F32_TFLOPS, 'traffic': traffic, 'traffic_source': note, 'launch_ms': acc[dom], 'patches_per_launch': B}
    split = acc.get(C5W_STAGE, 0.0) > 0
    shown = lambda k: 'conv5 backward, data path (dgrad + bn4 bwd sums)' if split and k == C5B_STAGE else k
    stages = {shown(k): {'ms': v, 'tflops': (2.0 * macs[k] * B / (v * 1e-3) / 1e12) if k in macs and v > 0 else None}
              for k, v in acc.items() if v > 0}
    return roofline, stages, len(stages), float(sum(acc.values()))


def measure_train(torch, dist, h, B, steps, warmup, rank, world, dev, st, profile):
    """`steps` training steps (scs_train_step + gradient all-reduce over RCCL when world > 1 + scs_adam_apply) of batch B per
    GPU between barrier + synchronize brackets, max over ranks."""
    inputs = train_inputs(torch, B, rank, dev)
    ax, co, sa, at, y = inputs
    ctr = [0]

    # one rank: the step applies the Adam update itself (scs_set_fused_adam - what Net.fit does without a process group: nothing
    # sits between the gradients and the update); N > 1: scs_train_step, the RCCL all-reduce, scs_adam_apply
    fused = not dist.is_dist()
    h.set_fused_adam(fused)

    def step():
        ctr[0] += 1
        h.train_step(ax, co, sa, at, y, B, seed=5 + ctr[0], want_loss=False, stream=st)
        if not fused:
            scale = dist.allreduce_grads(h, True, st, n_active=world)
            h.adam_apply(grad_scale=scale, stream=st)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    dt_own = time.perf_counter() - t0
    dt = dist.max_scalar(dt_own)
    value = world * B * steps / dt
    out = {'per_gpu_batch': B, 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * dt / steps, 'value': value, 'unit': 'patches/s',
           'net_tflops': value * TRAIN_FLOP / 1e12, 'net_frac_of_f32_peak': value * TRAIN_FLOP / 1e12 / (PEAK_F32_TFLOPS * world),
           'per_rank_ms_per_step': {'min': 1e3 * dist.min_scalar(dt_own) / steps, 'max': 1e3 * dt / steps}}
    # where a step's time goes outside scs_train_step (untimed extra steps; events on the stream the work is issued on, so that an
    # N > 1 line decomposes without a re-run): the gradient exchange (zero launches when N = 1) and the Adam launch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = [0.0, 0.0, 0.0]
    reps = 40
    h.set_fused_adam(False)                                            # the three parts as separate calls, whatever the timed form was
    for _ in range(reps):
        ctr[0] += 1
        ev[0].record()
        h.train_step(ax, co, sa, at, y, B, seed=5 + ctr[0], want_loss=False, stream=st)
        ev[1].record()
        scale = dist.allreduce_grads(h, True, st, n_active=world)
        ev[2].record()
        h.adam_apply(grad_scale=scale, stream=st)
        ev[3].record()
        torch.cuda.synchronize()
        for i in range(3):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
    out['train_step_ms'] = dist.max_scalar(acc[0] / reps)
    out['allreduce_ms'] = dist.max_scalar(acc[1] / reps)
    out['adam_ms'] = dist.max_scalar(acc[2] / reps)
    h.set_fused_adam(fused)
    out['update'] = 'fused into the step (one rank: scs_set_fused_adam)' if fused else 'scs_adam_apply behind the all-reduce'
    out['decomposition_note'] = ('%d extra steps with a synchronize after each (event pairs around scs_train_step / dist.allreduce_grads / '
                                 'scs_adam_apply, max over ranks): isolated steps, so train_step_ms exceeds the pipelined ms_per_step' % reps)
    if profile and rank == 0:
        out['roofline'], out['stages'], out['launches_per_step'], out['sum_of_launch_ms'] = train_stage_profile(h, inputs, B, st)
    return out, step


def make_cohort(root, n_subjects, seed, shape=(256, 256, 192), steered=False):
    """S synthetic subject folders of BASELINE configs[4] size on local disk: int16 T1, float32 15-channel prior (755 MB
    inflated), registered sub-cortical mask whose 10-voxel dilation gives a ~330 000-voxel ROI.  One subject is synthesised,
    the others are file copies of it (the work per subject does not depend on the voxel values).  steered: the prior is
    synth.steered_atlas - every one of the 15 classes wins somewhere in the volume (the `dice` block and the full-size test)."""
    import shutil
    from cnn_cort import nifti, synth
    first = os.path.join(root, 'subj000')
    os.makedirs(os.path.join(first, 'tmp'), exist_ok=True)
    t1 = synth.smooth_volume(shape, seed=seed, as_int=True)
    core = synth.ellipsoid_mask(shape, radii=(0.3, 0.27, 0.33))
    nifti.save(t1, os.path.join(first, 'T1.nii.gz'))
    prior = synth.steered_atlas(shape, seed=seed + 1)[0] if steered else synth.toy_atlas(shape, seed=seed + 1).astype(np.float32)
    nifti.save(prior, os.path.join(first, 'tmp', 'MNI_sub_probabilities.nii.gz'))
    del prior
    nifti.save(core.astype(np.uint8), os.path.join(first, 'tmp', 'MNI_subcortical_mask.nii.gz'))
    scans = [os.path.join(first, 'T1.nii.gz')]
    for i in range(1, n_subjects):
        sub = os.path.join(root, 'subj%03d' % i)
        shutil.copytree(first, sub)
        scans.append(os.path.join(sub, 'T1.nii.gz'))
    return scans


def cohort_mode(args, torch, dist, rank, world, local, weights):
    """BASELINE configs[4] end to end: base.test_cohort over S synthetic 256x256x192 subjects per rank, files on local disk,
    out_probabilities and post_process on (the reference's defaults) - the loop of train_model.py:74-78 with test_scan
    (base.py:401-458) inside.  value = subjects/hour over all ranks; the line also states what the host has to deliver for it:
    CPU-seconds per subject (process CPU time, all threads), GPU-seconds per subject, reader threads, usable CPUs, and how
    many such ranks one host share of that size could feed."""
    import shutil
    import tempfile
    from cnn_cort import base
    from cnn_cort.nets import build_model
    S = max(1, args.subjects)
    cpus = usable_cpus()
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    share = max(2, cpus // max(1, local_world))                       # host threads of this rank: readers, writers and the caller
    readers = args.readers or max(1, min(10, (share - 1) * 2 // 3))
    writers = args.writers or max(2, min(6, share - 1 - readers))
    root = tempfile.mkdtemp(prefix='scs_cohort_r%d_' % rank, dir=os.environ.get('SCS_BENCH_TMP'))
    try:
        t_make = time.perf_counter()
        scans = make_cohort(root, S + 1, seed=rank)                    # one extra subject for the warm-up
        t_make = time.perf_counter() - t_make
        opts = {'experiment': 'miccai2012_v1', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 20,
                'patch_size': [32, 32], 'load_weights': 'True', 'batch_size': 128, 'device': local, 'crop': 'True',
                'test_batch_size': 100000, 'out_probabilities': 'True', 'post_process': 'True', 'debug': 'False', 'towers': args.towers}
        net = build_model(os.path.join(PKG, 'nets'), opts)
        base.test_scan(net, scans[0], opts)                            # warm-up: library, workspace, page cache of the code paths
        n_roi = int(getattr(net, 'last_scan_voxels', 0))
        if hasattr(net, 'gpu_seconds'):
            net.gpu_seconds = 0.0
        torch.cuda.synchronize()
        dist.barrier()
        c0, t0 = time.process_time(), time.perf_counter()
        res = base.test_cohort(net, scans[1:], opts, shard=False, readers=readers, writers=writers)
        torch.cuda.synchronize()
        dist.barrier()
        dt = dist.max_scalar(time.perf_counter() - t0)
        cpu_s = dist.sum_scalar(time.process_time() - c0) / (world * S)
        gpu_s = dist.sum_scalar(float(getattr(net, 'gpu_seconds', 0.0))) / (world * S)
        assert len(res) == S
        value = world * S / dt * 3600.0
        per_rank_rate = S / dt                                         # subjects per second and rank
        out = {'metric': 'subjects/hour (test_cohort end to end: read, ROI, gather-fused inference, probabilities + post-processed labels written)',
               'value': value, 'unit': 'subjects/h', 'n_gpus': args.gpus, 'procs_per_gpu': world // args.gpus, 'ranks_seen': world,
               'steps': S, 'warmup': 1, 'ms_per_step': 1e3 * dt / S,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'whole-cohort inference, one subject at a time per GPU, %d synthetic 256x256x192 subjects per rank, '
                                      '~%d ROI voxels each, out_probabilities + post_process (BASELINE configs[4])' % (S, n_roi),
                          'weights': 'miccai2012_v1 (shipped)',
                          'parallelism': 'subjects sharded over %d rank(s) per GPU, no collective' % (world // args.gpus),
                          'towers': '%s (ran as: %s)' % (args.towers, getattr(net, 'last_scan_towers', '?'))},
               'host': {'cpu_seconds_per_subject': cpu_s, 'gpu_seconds_per_subject': gpu_s, 'wall_seconds_per_subject_per_rank': dt / S,
                        'usable_cpus': cpus, 'reader_threads_per_rank': readers, 'writer_threads_per_rank': writers,
                        'cores_busy_per_rank': per_rank_rate * cpu_s,
                        'ranks_one_host_share_can_feed': (cpus / (per_rank_rate * cpu_s)) if cpu_s > 0 else None,
                        'synth_seconds': t_make},
               'gpu_stage_voxels_per_s': (n_roi / gpu_s) if gpu_s > 0 else None}
        if rank == 0 and world == 1 and not args.no_cpu:
            cb = cpu_baseline_volume(weights)
            per_subject = (n_roi or 400000) / cb['value']
            out['cpu_baseline'] = {'value': 3600.0 / per_subject, 'unit': 'subjects/h', 'cores': cb['cores'], 'kind': 'port',
                                   'sample': cb['sample'] + ' - GPU stage of one subject only; the file stages are the same code on both sides'}
        if rank == 0:
            print(json.dumps(out, default=float))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def sustained_block(torch, dist, fwd_step, B, burst_ms, train_step, train_B, train_burst_ms, world):
    """The same steps again, for seconds instead of a burst (the contract's timed region is 20 x 35 ms): >= 8 s of forward
    steps and >= 3 s of training steps, each between barrier + synchronize brackets, max over ranks.  The step counts come from
    the (already max-reduced) burst times, so every rank runs the same number."""
    out = {}
    for name, step, b, ms, secs in (('forward', fwd_step, B, burst_ms, 8.0), ('train_batch%d' % train_B, train_step, train_B, train_burst_ms, 3.0)):
        if step is None or not ms:
            continue
        k = int(secs * 1e3 / ms) + 1
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        dt = dist.max_scalar(time.perf_counter() - t0)
        rate, burst = world * b * k / dt, world * b / (ms * 1e-3)
        out[name] = {'steps': k, 'seconds': dt, 'value': rate, 'unit': 'patches/s', 'burst_value': burst, 'ratio_to_burst': rate / burst}
    return out


def dice_block(local, weights, n_compare=24000):
    """Per-class Dice of the GPU labels against the CPU port (BASELINE.json `metric`: "...; per-class Dice vs CPU ref"): one
    synthetic 256x256x192 subject whose prior makes every class win somewhere (synth.steered_atlas) through build_model +
    test_scan (base.py:401-458; raw labels and probabilities written), then `n_compare` of its ROI voxels - every voxel the GPU
    labelled 1..14 up to a cap per class, the rest drawn from the whole ROI - through the float32 torch-CPU port on patches cut
    from the NumPy-normalised volume (base.py:357-358, 293-306).  No real T1 / atlas ships with the reference: the
    figure is label-for-label agreement of the two implementations, not anatomical accuracy."""
    import shutil
    import tempfile
    import torch
    from cnn_cort import base, nifti
    from cnn_cort.nets import build_model
    from oracle import patches_ref as R, torch_ref as T
    root = tempfile.mkdtemp(prefix='scs_dice_', dir=os.environ.get('SCS_BENCH_TMP'))
    try:
        t0 = time.perf_counter()
        scan = make_cohort(root, 1, seed=3, steered=True)[0]
        sub = os.path.dirname(scan)
        t_make = time.perf_counter() - t0
        opts = {'experiment': 'miccai2012_v1', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 20,
                'patch_size': [32, 32], 'load_weights': 'True', 'batch_size': 128, 'device': local, 'crop': 'True',
                'test_batch_size': 100000, 'out_probabilities': 'True', 'post_process': 'False', 'debug': 'False'}
        net = build_model(os.path.join(PKG, 'nets'), opts)
        t0 = time.perf_counter()
        base.test_scan(net, scan, opts)
        t_scan = time.perf_counter() - t0
        seg = nifti.load(os.path.join(sub, 'out_subcortical_rawseg.nii.gz')).get_data()
        prob = nifti.load(os.path.join(sub, 'out_subcortical_prob.nii.gz')).get_data()
        n_roi = int(getattr(net, 'last_scan_voxels', 0))
        written = np.argwhere(prob.sum(-1) != 0)                          # the ROI voxels (every row of probabilities sums to 1)
        rng = np.random.RandomState(0)
        lab_w = seg[written[:, 0], written[:, 1], written[:, 2]]
        pick = set(rng.choice(len(written), min(len(written), n_compare // 2), replace=False).tolist())
        for c in range(1, 15):
            idx = np.flatnonzero(lab_w == c)
            pick.update(rng.choice(idx, min(len(idx), n_compare // 28), replace=False).tolist())
        c = written[np.sort(np.fromiter(pick, dtype=np.int64))]
        raw = nifti.load(scan).get_data()
        vn = R.zscore_inference(raw).astype(np.float32)
        atlas = nifti.load(os.path.join(sub, 'tmp', 'MNI_sub_probabilities.nii.gz')).get_data()
        rows = atlas[c[:, 0], c[:, 1], c[:, 2]].astype(np.float32)
        zero = rows.sum(1) == 0
        rows[zero, 14] = 1.0                                               # base.py:391-394
        del atlas
        cores = usable_cpus()
        torch.set_num_threads(cores)
        cpu = T.FoldedCPUNet(weights)
        t0 = time.perf_counter()
        p_cpu = []
        for lo in range(0, len(c), 2048):
            cc = c[lo:lo + 2048]
            views = [torch.from_numpy(np.ascontiguousarray(base.get_patches(vn, cc, (32, 32), m)[:, None])) for m in ('axial', 'coronal', 'saggital')]
            p_cpu.append(cpu.forward(views[0], views[1], views[2], torch.from_numpy(rows[lo:lo + 2048])).numpy())
        p_cpu = np.concatenate(p_cpu)
        t_cpu = time.perf_counter() - t0
        l_cpu = p_cpu.argmax(1)
        l_gpu = seg[c[:, 0], c[:, 1], c[:, 2]].astype(np.int64)
        p_gpu = prob[c[:, 0], c[:, 1], c[:, 2]]
        srt = np.sort(p_cpu, axis=1)
        per_class, seen = {}, 0
        for k in range(15):
            a, b = l_gpu == k, l_cpu == k
            if a.sum() + b.sum():
                seen += 1
                per_class[str(k)] = {'dice': 2.0 * float((a & b).sum()) / float(a.sum() + b.sum()), 'gpu_voxels': int(a.sum()), 'cpu_voxels': int(b.sum())}
        differ = np.flatnonzero(l_gpu != l_cpu)
        # the same subject through test_scan on the SPLIT kernels (options fwd_split + per-voxel towers: the shared-tower evaluation
        # stays on the f32 kernels), compared with the same CPU results and with the f32 path's labels over the whole ROI
        split = None
        try:
            o2 = dict(opts, fwd_split='True', towers='patch')
            net2 = build_model(os.path.join(PKG, 'nets'), o2)
            t0 = time.perf_counter()
            base.test_scan(net2, scan, o2)
            t_scan2 = time.perf_counter() - t0
            seg2 = nifti.load(os.path.join(sub, 'out_subcortical_rawseg.nii.gz')).get_data()
            prob2 = nifti.load(os.path.join(sub, 'out_subcortical_prob.nii.gz')).get_data()
            l2 = seg2[c[:, 0], c[:, 1], c[:, 2]].astype(np.int64)
            pc2 = {}
            for k in range(15):
                a, b = l2 == k, l_cpu == k
                if a.sum() + b.sum():
                    pc2[str(k)] = 2.0 * float((a & b).sum()) / float(a.sum() + b.sum())
            split = {'options': "fwd_split='True', towers='patch' (f16 x 2 pieces on the f16 matrix cores, f32 accumulate)",
                     'towers': net2.last_scan_towers, 'split_fallbacks': int(net2._handle.get_option('split_fallbacks')),
                     'min_dice': min(pc2.values()), 'classes_seen': len(pc2), 'labels_differ_vs_cpu': int((l2 != l_cpu).sum()),
                     'labels_differ_vs_f32_whole_roi': int((seg2 != seg).sum()),
                     'max_abs_prob_diff_vs_cpu': float(np.abs(prob2[c[:, 0], c[:, 1], c[:, 2]] - p_cpu).max()),
                     'max_abs_prob_diff_vs_f32': float(np.abs(prob2[c[:, 0], c[:, 1], c[:, 2]] - p_gpu).max()),
                     'test_scan_seconds': t_scan2}
        except Exception as e:                                             # the f32 block above is the metric; say why this one is missing
            split = {'error': repr(e)}
        return {'subject': 'synthetic 256x256x192 int16 T1, steered 15-channel prior (cnn_cort.synth.steered_atlas), %d ROI voxels' % n_roi,
                'reference': 'float32 torch-CPU port (oracle/torch_ref.py) on %d host threads - kind "port": the Theano path cannot run here' % cores,
                'voxels_compared': int(len(c)), 'classes_seen': seen, 'classes_in_volume': int(len(np.unique(seg))),
                'per_class': per_class, 'min_dice': min(v['dice'] for v in per_class.values()),
                'labels_differ': int(len(differ)),
                'labels_differ_max_cpu_margin': float((srt[differ, -1] - srt[differ, -2]).max()) if len(differ) else 0.0,
                'max_abs_prob_diff': float(np.abs(p_gpu - p_cpu).max()), 'split': split,
                'seconds': {'synthesise_and_write_subject': t_make, 'test_scan_gpu': t_scan, 'cpu_port': t_cpu}}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no torchrun environment: start N ranks as CHILD processes (one per GPU,
    torch.distributed.run, rendezvous on 127.0.0.1) and exit with their code.  Called before anything in this process
    touches the GPU (importing torch does not; counting devices does not initialise the runtime on this image)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # RCCL needs dmabuf IPC on this pool
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', choices=['forward', 'train', 'volume', 'cohort'], default='forward')
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-train-block', action='store_true', help='forward mode: skip the train_step block')
    ap.add_argument('--no-sustained', action='store_true', help='forward mode: skip the sustained block (>= 8 s forward, >= 3 s train)')
    ap.add_argument('--dtype', choices=['f32', 'f16x2'], default='f32',
                    help='forward mode: f32 = the exact-f32 MFMA kernels (the headline line); f16x2 = the split kernels (option fwd_split) as '
                         'the line\'s value. The default line carries the split path as its "split" block either way')
    ap.add_argument('--no-split', action='store_true', help='forward mode: skip the split block')
    ap.add_argument('--no-dice', action='store_true', help='forward mode: skip the dice block (full-size subject, GPU labels vs CPU port)')
    ap.add_argument('--train-steps', type=int, default=200, help='timed steps per batch size of the train_step block')
    ap.add_argument('--subjects', type=int, default=6, help='cohort mode: synthetic subjects per rank')
    ap.add_argument('--procs-per-gpu', type=int, default=1,
                    help='cohort mode: ranks that share one GPU (gloo; the host side of a subject is Python threads, one process cannot '
                         'keep more than ~6 cores busy)')
    ap.add_argument('--writers', type=int, default=0, help='cohort mode: writer threads per rank (0 = from the usable CPUs)')
    ap.add_argument('--towers', choices=['auto', 'patch', 'shared'], default='auto',
                    help="cohort mode: 'patch' = one tower evaluation per ROI voxel and view (the reference's procedure), 'shared' = tower "
                         "work shared between neighbouring voxels (same bits), 'auto' = shared for dense ROIs")
    ap.add_argument('--readers', type=int, default=0, help='cohort mode: reader threads per rank (0 = from the usable CPUs / ranks on this host, at most 10)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher rehearsal without a GPU: ranks rendezvous (gloo), run the barrier / max-over-ranks '
                         'timing brackets around an empty step and rank 0 prints a line marked "dry_run": true, value null')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    per_gpu = max(1, args.procs_per_gpu) if args.mode == 'cohort' else 1
    want_world = args.gpus * per_gpu                                  # ranks of the job (= --gpus except for a shared-GPU cohort run)
    if want_world > 1 and 'WORLD_SIZE' not in os.environ:
        if per_gpu > 1:
            os.environ['SCS_DIST_BACKEND'] = 'gloo'                    # several ranks per GPU: no RCCL communicator (and no collective on the data path)
        raise SystemExit(spawn_ranks(want_world, sys.argv[1:]))
    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    if env_world != want_world:
        raise SystemExit('bench.py: --gpus %d (x %d) but the launcher started %d rank(s); refusing to report a run whose '
                         'world size differs from what was asked' % (args.gpus, per_gpu, env_world))

    import torch
    from cnn_cort import dist

    if args.dry_run:
        rank, world, _ = dist.init_from_env(backend='gloo')
        assert world == want_world
        ranks_seen = int(round(dist.sum_scalar(1.0)))
        dist.barrier()
        t0 = time.perf_counter()
        dist.barrier()
        dt = dist.max_scalar(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({'metric': 'tri-view 32x32 patches/s (%s)' % args.mode, 'value': None, 'unit': 'patches/s',
                              'n_gpus': world, 'ranks_seen': ranks_seen, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
                              'dry_run': True, 'bracket_s': dt}))
        return

    from cnn_cort import _hip, params as P

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device')
    if torch.cuda.device_count() < args.gpus and not os.environ.get('SCS_DIST_BACKEND'):
        raise SystemExit('bench.py: --gpus %d but only %d HIP device(s) are visible (set SCS_DIST_BACKEND=gloo for a '
                         'shared-GPU rehearsal)' % (args.gpus, torch.cuda.device_count()))
    rank, world, local = dist.init_from_env()
    # every rank checks the communicator it actually got, and rank 0 reports how many ranks answered a collective
    if world != want_world or dist.world_size() != want_world:
        raise SystemExit('bench.py: rank %d sees a process group of %d rank(s), %d were asked for' % (rank, dist.world_size(), want_world))
    ranks_seen = int(round(dist.sum_scalar(1.0)))
    if ranks_seen != want_world:
        raise SystemExit('bench.py: a sum over the ranks counted %d, %d were asked for' % (ranks_seen, want_world))
    torch.cuda.set_device(local)
    dev = 'cuda:%d' % local
    weights = P.load_params(os.path.join(PKG, 'nets', 'miccai2012_v1', 'miccai2012_v1'))
    if args.mode == 'cohort':
        return cohort_mode(args, torch, dist, rank, world, local, weights)
    h = _hip.Handle(local)
    h.set_params(P.params_to_flat(weights))
    if args.chunk:
        h.set_chunk(args.chunk)
    chunk = h.set_chunk(0)
    st = torch.cuda.current_stream().cuda_stream
    backend = torch.distributed.get_backend() if dist.is_dist() else None

    train_out = None
    if args.mode == 'train':
        B = args.batch or 256
        if args.dtype == 'f16x2':
            h.set_option('train_split', 1)
        train_out, train_step_fn = measure_train(torch, dist, h, B, args.steps, args.warmup, rank, world, dev, st, profile=True)
        value, dt_ms = train_out['value'], train_out['ms_per_step']
        flop_per_patch = TRAIN_FLOP
        workload = 'training step (fwd+bwd+Adam), synthetic batch %d per GPU, 15 classes (BASELINE configs[2]/[3])' % B
    else:
        if args.mode == 'forward':
            B = args.batch or 100000
            ax, co, sa, at = synth_on_device(torch, B, 1000 * rank, dev)
            probs = torch.empty((B, 15), device=dev)
            labels = torch.empty((B,), dtype=torch.int32, device=dev)

            def step():
                h.forward(ax, co, sa, at, B, probs, labels, st)
            if args.dtype == 'f16x2':
                h.set_option('fwd_split', 1)
            workload = 'forward-only inference, synthetic 32x32x3-view float32 patches, batch %d per GPU (BASELINE configs[1])' % B
        else:
            # BASELINE configs[4], GPU stage only: one subject per GPU, gather-fused inference over a ~400 000-voxel ROI of a
            # 256x256x192 volume (synthetic, z-scored noise), labels + probabilities from one pass (--mode cohort times the
            # whole of test_scan, files included)
            dims = (256, 256, 192)
            g = torch.Generator(device=dev)
            g.manual_seed(7 + rank)
            vol = torch.randn(dims, generator=g, device=dev, dtype=torch.float32)
            e = torch.rand(dims + (15,), generator=g, device=dev, dtype=torch.float32)
            atlas4d = (e / e.sum(-1, keepdim=True)).contiguous()
            zz, yy, xx = torch.meshgrid(*[torch.arange(s, device=dev) for s in dims], indexing='ij')
            roi = (((zz - 128) / 60.0) ** 2 + ((yy - 128) / 52.0) ** 2 + ((xx - 96) / 31.0) ** 2) <= 1.0
            centers = torch.nonzero(roi).to(torch.int32).contiguous()          # C order, like np.nonzero (base.py:310-331)
            B = int(centers.shape[0])
            probs = torch.empty((B, 15), device=dev)
            labels = torch.empty((B,), dtype=torch.int32, device=dev)

            def step():
                h.forward_volume(vol, dims, centers, atlas4d, B, probs, labels, st)
            workload = 'gather-fused whole-subject inference (scs_forward_volume), 256x256x192 volume, %d ROI voxels per GPU (BASELINE configs[4], GPU stage)' % B
        flop_per_patch = FWD_FLOP
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        dt_own = time.perf_counter() - t0
        dt = dist.max_scalar(dt_own)
        value = world * B * args.steps / dt
        dt_ms = 1e3 * dt / args.steps
        rank_spread = {'min': 1e3 * dist.min_scalar(dt_own) / args.steps, 'max': dt_ms}

    out = {
        'metric': 'tri-view 32x32 patches/s (%s)' % {'forward': 'forward', 'train': 'train step', 'volume': 'forward, gather-fused'}[args.mode],
        'value': value, 'unit': 'patches/s', 'n_gpus': world, 'ranks_seen': ranks_seen, 'dist_backend': backend,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt_ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.dtype == 'f32' or args.mode == 'volume' else (SPLIT_DTYPE if args.mode == 'forward' else TRAIN_SPLIT_DTYPE), 'data': 'synthetic',
        'config': {'workload': workload, 'weights': 'miccai2012_v1 (shipped)', 'per_gpu_batch': B, 'chunk': chunk,
                   'parallelism': 'independent patch shards, no data-path collective' if args.mode != 'train'
                   else 'dp%d, one RCCL all-reduce of the 3.5 MB gradient arena per step' % world,
                   **collective_config(world)},
        'net_tflops': value * flop_per_patch / 1e12,
        'net_frac_of_f32_peak': value * flop_per_patch / 1e12 / (PEAK_F32_TFLOPS * world),
    }
    if args.mode != 'train':
        out['per_rank_ms_per_step'] = rank_spread

    if rank == 0 and args.mode == 'forward':
        # dominant kernel, timed with HIP events on the launch stream (untimed region)
        ms, launches = h.forward_profiled(ax, co, sa, at, B, probs, labels, st)
        ms2, _ = h.forward_profiled(ax, co, sa, at, B, probs, labels, st)
        ms = 0.5 * (ms + ms2)                                             # average launch duration over both passes
        names, macs = list(STAGE_NAMES), list(STAGE_MACS)
        if ms[2] < 0.1 * ms[1] and ms[3] < 0.1 * ms[1]:
            # conv3 -> conv4 + pool2 -> conv5 ran as ONE kernel per patch-tower (csrc/conv_fused.h: the profiled forward books it
            # under the first of the three stages): one stage with the three layers' work
            names[1], macs[1] = 'conv3+conv4+pool2+conv5 (one kernel)', STAGE_MACS[1] + STAGE_MACS[2] + STAGE_MACS[3]
            macs[2] = macs[3] = 0
        dom = int(np.argmax(ms))
        per_launch_patches = B / launches
        flop_launch = 2.0 * macs[dom] * per_launch_patches
        ms = ms.astype(np.float64)
        dur = float(ms[dom]) / launches * 1e-3
        achieved = flop_launch / dur / 1e12
        traffic, note = measured_traffic(names[dom]) if B == 100000 else (None, 'measured at the default batch only')
        out['roofline'] = {'bound': 'mfma', 'kernel': names[dom], 'achieved': achieved, 'peak': PEAK_F32_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_TFLOPS, 'traffic': traffic, 'traffic_source': note,
                           'launch_ms': dur * 1e3, 'patches_per_launch': per_launch_patches}
        # beside the nominal peak: what this device issues when it does nothing but the MFMA the kernels are built on (live probe)
        try:
            issue, ghz = h.probe_f32_mfma_clock()
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            out['roofline']['device_mfma_issue_rate'] = {
                'tflops': issue, 'shader_clock_ghz': ghz,
                # 4 SIMDs per CU x 2048 FLOP per instruction / 32 cycles at the measured clock: what the pipe gives at THAT clock
                'pipe_rate_at_that_clock_tflops': cus * 4 * 2048 / 32 * ghz * 1e9 / 1e12 if ghz else None,
                'what': 'back-to-back v_mfma_f32_16x16x4_f32, independent accumulators, operands in registers, four waves per SIMD on '
                        'every CU (scs_probe_f32_mfma_clock, three launches of ~4 ms, the last one timed, this process), and the shader clock the '
                        'device held meanwhile (s_memtime / s_memrealtime x 100 MHz, median over the workgroups; constant operands: the least power '
                        'an MFMA loop can draw).  The nominal peak prices the instruction at 32 cycles of 2.4 GHz; tflops / pipe_rate_at_that_clock '
                        'is the share of the pipe the loop reaches in cycles.  Informational: `frac` above is against the nominal peak'}
        except Exception as e:                                            # informational: never fails the bench line
            out['roofline']['device_mfma_issue_rate'] = {'error': str(e)}
        out['stages'] = {names[i]: {'ms_per_step': float(ms[i]),
                                    'tflops': float(2.0 * macs[i] * B / (ms[i] * 1e-3) / 1e12) if ms[i] > 1e-3 else None}
                         for i in range(8) if macs[i] > 0}
        out['hbm_algorithmic_gbps'] = value / world * ALGO_BYTES_FWD / 1e9
    if args.mode == 'forward' and args.dtype == 'f32' and not args.no_split:
        # (an extra block: whatever goes wrong in it must not cost the line its headline - the error is reported instead)
        try:
            # the same workload on the split kernels (csrc/conv_split.h), same contract (warm-up, K steps between barriers, max over
            # ranks), its results compared with the f32 path's on the same inputs
            p32, l32 = probs.clone(), labels.clone()
            h.set_option('fwd_split', 1)
            h.set_option('split_fallbacks', 0)
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dist.barrier()
            sdt = dist.max_scalar(time.perf_counter() - t0)
            sval = world * B * args.steps / sdt
            if rank == 0:
                sms, sl = h.forward_profiled(ax, co, sa, at, B, probs, labels, st)
                sms2, _ = h.forward_profiled(ax, co, sa, at, B, probs, labels, st)
                sms = (0.5 * (sms + sms2)).astype(np.float64)
                sdom = int(np.argmax(sms))
                sdur = float(sms[sdom]) / sl * 1e-3
                sflop = 2.0 * STAGE_MACS[sdom] * B / sl
                diff = (probs - p32).abs().max().item()
                nd = int((labels != l32).sum().item())
                srt = torch.sort(p32, dim=1).values
                margin = (srt[:, -1] - srt[:, -2])
                out['split'] = {
                    'value': sval, 'unit': 'patches/s', 'ms_per_step': 1e3 * sdt / args.steps, 'steps': args.steps, 'warmup': args.warmup,
                    'ratio_to_f32_line': sval / value, 'dtype': SPLIT_DTYPE,
                    'net_f32_equivalent_tflops': sval * FWD_FLOP / 1e12,
                    'vs_f32_path': {'max_abs_prob_diff': diff, 'labels_differ': nd,
                                    'labels_differ_max_f32_margin': float(margin[labels != l32].max().item()) if nd else 0.0,
                                    'batches_repeated_on_f32_kernels': h.get_option('split_fallbacks')},
                    'roofline': {'bound': 'mfma', 'kernel': STAGE_NAMES[sdom],
                                 'achieved': 3.0 * sflop / sdur / 1e12, 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s',
                                 'frac': 3.0 * sflop / sdur / 1e12 / PEAK_F16_TFLOPS,
                                 'achieved_f32_equivalent': sflop / sdur / 1e12,
                                 'traffic': measured_traffic('split: ' + STAGE_NAMES[sdom])[0] if B == 100000 else None,
                                 'note': 'algorithmic work of the split form = three f16 products per f32 multiply-add (padding of the 16-wide channel tiles '
                                         'and 32-deep k blocks not counted); priced against the dense f16 MFMA peak',
                                 'launch_ms': sdur * 1e3, 'patches_per_launch': B / sl},
                    'stages': {STAGE_NAMES[i]: {'ms_per_step': float(sms[i]),
                                                'f32_equivalent_tflops': float(2.0 * STAGE_MACS[i] * B / (sms[i] * 1e-3) / 1e12) if sms[i] > 1e-3 else None}
                               for i in range(8)},
                }
        except Exception as e:                                              # noqa: BLE001
            if rank == 0:
                out['split'] = {'error': '%s: %s' % (type(e).__name__, e)}
        h.set_option('fwd_split', -1)
        step()                                                              # probs / labels of the f32 path again
        torch.cuda.synchronize()
    if args.mode == 'train' and rank == 0:
        for k in ('roofline', 'stages', 'launches_per_step', 'sum_of_launch_ms', 'train_step_ms', 'allreduce_ms', 'adam_ms',
                  'decomposition_note', 'per_rank_ms_per_step', 'update'):
            out[k] = train_out[k]
    if args.mode == 'forward':
        # the other half of BASELINE.json's metric, in the same driver-run line: the training step at batch 256 (configs[2])
        # and 128 (nolearn's effective batch in the reference), every rank takes part (the step has a collective when N > 1)
        blk, tstep, tms = {}, None, None
        if not args.no_train_block:
            for tb in (128, 256):
                r, stp = measure_train(torch, dist, h, tb, args.train_steps, 20, rank, world, dev, st, profile=True)
                r['frac_of_f32_peak'] = r['net_frac_of_f32_peak']
                issue = (out.get('roofline') or {}).get('device_mfma_issue_rate', {}).get('tflops')
                if issue:                                                 # beside the nominal peak, as in the forward's roofline block
                    r['net_frac_of_device_mfma_issue_rate'] = r['net_tflops'] / (issue * world)
                blk['batch%d' % tb] = r
                tstep, tms = stp, r['ms_per_step']
            sblk = {}
            if not args.no_split:
                # the same steps with option train_split (csrc/conv_split_train.h): same contract, same batches
                try:
                    h.set_option('train_split', 1)
                    for tb in (128, 256):
                        r, _ = measure_train(torch, dist, h, tb, args.train_steps, 20, rank, world, dev, st, profile=True)
                        r['frac_of_f32_peak'] = r['net_frac_of_f32_peak']
                        r['ratio_to_f32_step'] = blk['batch%d' % tb]['ms_per_step'] / r['ms_per_step']
                        sblk['batch%d' % tb] = r
                    sblk['dtype'] = TRAIN_SPLIT_DTYPE
                except Exception as e:                                      # noqa: BLE001  (an extra block must not cost the line)
                    sblk = {'error': '%s: %s' % (type(e).__name__, e)}
                h.set_option('train_split', -1)
            if rank == 0:
                out['train_step'] = dict(blk, metric='tri-view 32x32 patches/s (train step: scs_train_step + all-reduce + scs_adam_apply)',
                                         flop_per_patch=TRAIN_FLOP, dtype='f32', data='synthetic', collective=collective_config(world))
                if sblk:
                    out['train_step']['split'] = sblk
        if not args.no_sustained:
            sus = sustained_block(torch, dist, step, B, dt_ms, tstep, 256, tms, world)
            if rank == 0:
                out['sustained'] = sus
        del ax, co, sa, at, probs, labels
    if args.mode == 'train' and not args.no_sustained:
        k = int(3.0e3 / dt_ms) + 1
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            train_step_fn()
        torch.cuda.synchronize()
        dist.barrier()
        sdt = dist.max_scalar(time.perf_counter() - t0)
        if rank == 0:
            out['sustained'] = {'steps': k, 'seconds': sdt, 'value': world * B * k / sdt, 'unit': 'patches/s', 'burst_value': value,
                                'ratio_to_burst': world * B * k / sdt / value}
    if args.mode == 'forward' and rank == 0 and world == 1 and not args.no_dice:
        out['dice'] = dice_block(local, weights)
    if rank == 0 and world == 1 and not args.no_cpu:
        out['cpu_baseline'] = {'train': cpu_baseline_train, 'volume': cpu_baseline_volume, 'forward': cpu_baseline}[args.mode](weights)
    if rank == 0:
        print(json.dumps(out, default=float))
    h.close()


if __name__ == '__main__':
    main()
