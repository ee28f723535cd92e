"""Does replaying the training step as a HIP graph beat issuing its 27 launches?  One step (scs_train_step + scs_adam_apply,
batch 256 and 128) captured with torch.cuda.CUDAGraph on a side stream, replayed, timed against the plain calls.
python tools/graph_train_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
sys.path.insert(0, ROOT)

import torch

import bench
from cnn_cort import _hip, params as P


def main():
    weights = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))
    for B in (256, 128):
        h = _hip.Handle(0)
        h.set_params(P.params_to_flat(weights))
        fused = os.environ.get('SCS_ISSUE_FUSED', '1') != '0'       # the update inside the step, as bench.py runs one rank (a probe: a
        if fused:                                                    # replayed graph repeats the captured step's Adam coefficients)
            h.set_fused_adam(True)
        ax, co, sa, at, y = bench.train_inputs(torch, B, 0, 'cuda:0')
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            st = s.cuda_stream

            def step(seed):
                h.train_step(ax, co, sa, at, y, B, seed=seed, want_loss=False, stream=st)
                if not fused:
                    h.adam_apply(stream=st)
            for i in range(20):
                step(5 + i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(300):
                step(5)
            torch.cuda.synchronize()
            plain = (time.perf_counter() - t0) / 300
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, stream=s):
                    step(5)
            except Exception as e:            # noqa: BLE001 - a probe: say what refused
                print('batch %d: capture failed: %r' % (B, e))
                continue
            torch.cuda.synchronize()
            for i in range(20):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(300):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / 300
        print('batch %d: %.4f ms per step issued launch by launch, %.4f ms replayed as a graph' % (B, plain * 1e3, graph * 1e3))
        h.close()


if __name__ == '__main__':
    main()
