"""Experiment: how much of the training step at small batch is launch / dispatch overhead?  Captures one
scs_train_step + scs_adam_apply into a HIP graph (fixed inputs, fixed dropout seed, frozen Adam step size: timing only)
and replays it, against the same step issued kernel by kernel.      python tools/graph_probe.py [batch]
(one batch size per process: a second capture on the same handle is not supported by this probe)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cnn_cort import _hip, params as P, synth

w = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
h = _hip.Handle(0)
h.set_params(P.params_to_flat(w))
for B in [int(a) for a in sys.argv[1:2]] or [128]:
    ax, co, sa, at = [torch.from_numpy(a).cuda() for a in synth.patch_batch(B, seed=0)]
    y = torch.from_numpy(np.random.RandomState(4).randint(0, 15, B).astype(np.uint8)).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def step():
            h.train_step(ax, co, sa, at, y, B, seed=5, want_loss=False, stream=s.cuda_stream)
            h.adam_apply(stream=s.cuda_stream)
        for _ in range(5):
            step()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        s.synchronize()
        t_plain = (time.perf_counter() - t0) / 100
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(5):
            g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            g.replay()
        s.synchronize()
        t_graph = (time.perf_counter() - t0) / 100
    print('batch %4d: kernel-by-kernel %.3f ms/step, graph replay %.3f ms/step' % (B, t_plain * 1e3, t_graph * 1e3))
