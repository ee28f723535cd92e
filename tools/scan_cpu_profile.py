"""CPU-seconds (process time, all threads) and wall time per stage of cnn_cort.base.test_scan on one synthetic subject of
BASELINE configs[4] size - what a host share has to deliver per subject.   python tools/scan_cpu_profile.py [--no-gpu]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
sys.path.insert(0, ROOT)

import numpy as np

import bench
from cnn_cort import base, nifti


class Meter:
    def __init__(self):
        self.rows = {}

    def wrap(self, obj, name, label):
        fn = getattr(obj, name)

        def timed(*a, **k):
            c0, t0 = time.process_time(), time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                r = self.rows.setdefault(label, [0.0, 0.0])
                r[0] += time.process_time() - c0
                r[1] += time.perf_counter() - t0
        setattr(obj, name, timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-gpu', action='store_true')
    ap.add_argument('--repeat', type=int, default=3)
    args = ap.parse_args()
    opts = {'experiment': 'miccai2012_v1', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 20,
            'patch_size': [32, 32], 'load_weights': 'True', 'batch_size': 128, 'device': 0, 'crop': 'True',
            'test_batch_size': 100000, 'out_probabilities': 'True', 'post_process': 'True', 'debug': 'False'}
    if args.no_gpu:
        class StubNet:
            def predict_volume(self, vol, centers, atlas4d=None, want_probs=True, batch=None, atlas_rows=None, towers='auto'):
                n = len(centers)
                rng = np.random.RandomState(0)
                p = rng.dirichlet(np.ones(15), n).astype(np.float32)
                return p.argmax(1).astype(np.int64), (p if want_probs else None)
        net = StubNet()
    else:
        from cnn_cort.nets import build_model
        net = build_model(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets'), opts)
    m = Meter()
    for obj, name, label in ((nifti, 'load', 'nifti.load (T1, mask)'), (nifti, 'save', 'nifti.save (labels)'),
                             (base, '_zscore', 'z-score'), (base, 'get_mask_voxels', 'ROI voxel list'),
                             (nifti, 'load_rows', 'prior rows (inflate of the atlas)'), (nifti, 'save_sparse', 'probability file'),
                             (base, 'post_process_segmentation', 'post_process'), (net, 'predict_volume', 'GPU path (upload, forward, download)')):
        if hasattr(obj, name):
            m.wrap(obj, name, label)
    if hasattr(net, 'dilate_mask'):
        m.wrap(net, 'dilate_mask', 'ROI dilation x10 (GPU)')
    with tempfile.TemporaryDirectory() as d:
        scans = bench.make_cohort(d, 1, seed=0)
        for r in range(args.repeat):
            m.rows.clear()
            c0, t0 = time.process_time(), time.perf_counter()
            base.test_scan(net, scans[0], opts)
            cpu, wall = time.process_time() - c0, time.perf_counter() - t0
            print('--- test_scan run %d: wall %.3f s, CPU %.3f s' % (r, wall, cpu))
            for k, (c, t) in sorted(m.rows.items(), key=lambda kv: -kv[1][0]):
                print('    %-45s CPU %6.3f s   wall %6.3f s' % (k, c, t))
            print('    %-45s CPU %6.3f s   wall %6.3f s' % ('everything else', cpu - sum(c for c, _ in m.rows.values()), wall - sum(t for _, t in m.rows.values())))


if __name__ == '__main__':
    main()
