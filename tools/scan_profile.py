"""
Stage timing of cnn_cort.base.test_scan on one synthetic subject of BASELINE configs[4] size (256x256x192 T1,
15-channel atlas, ~400k-voxel ROI after the 10-iteration dilation): where a subject's wall-clock goes between
disk, host preparation, the GPU path and the output files.

    python tools/scan_profile.py [--shape 256 256 192] [--no-gpu]      (writes into a temp dir, prints a table)

--no-gpu replaces net.predict_volume by a stub (host stages only; for the CPU container).
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
sys.path.insert(0, ROOT)

import numpy as np
from scipy import ndimage

from cnn_cort import base, nifti, synth


class Stopwatch:
    def __init__(self):
        self.t = {}

    def wrap(self, obj, name, label=None):
        fn = getattr(obj, name)
        label = label or name

        def timed(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                self.t[label] = self.t.get(label, 0.0) + time.perf_counter() - t0
        setattr(obj, name, timed)


def make_subject(folder, shape, seed=0):
    os.makedirs(os.path.join(folder, 'tmp'), exist_ok=True)
    t1 = synth.smooth_volume(shape, seed=seed, as_int=True)
    # small core: the reference dilates the atlas mask by 10 voxels to get the ROI (base.py:369)
    core = synth.ellipsoid_mask(shape, radii=(0.3, 0.27, 0.33))
    atlas = synth.toy_atlas(shape, seed=seed + 1)
    nifti.save(t1, os.path.join(folder, 'T1.nii.gz'))
    nifti.save(atlas.astype(np.float32), os.path.join(folder, 'tmp', 'MNI_sub_probabilities.nii.gz'))
    nifti.save(core.astype(np.uint8), os.path.join(folder, 'tmp', 'MNI_subcortical_mask.nii.gz'))
    return os.path.join(folder, 'T1.nii.gz')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', type=int, nargs=3, default=[256, 256, 192])
    ap.add_argument('--no-gpu', action='store_true')
    ap.add_argument('--repeat', type=int, default=2)
    ap.add_argument('--post', action='store_true', help='also run post_process (largest-overlap component filter)')
    ap.add_argument('--cohort', type=int, default=0, help='also time test_cohort (pipelined) against the test_scan loop on N subjects')
    args = ap.parse_args()
    shape = tuple(args.shape)
    opts = {'experiment': 'miccai2012_v1', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 20,
            'patch_size': [32, 32], 'load_weights': 'True', 'batch_size': 128, 'device': 0, 'crop': True,
            'test_batch_size': 100000, 'out_probabilities': True, 'post_process': args.post, 'debug': False}

    if args.no_gpu:
        class StubNet:
            def predict_volume(self, vol, centers, atlas4d=None, want_probs=True, batch=None, atlas_rows=None, towers='auto'):
                n = len(centers)
                return np.zeros(n, np.int32), (np.zeros((n, 15), np.float32) if want_probs else None)
        net = StubNet()
    else:
        from cnn_cort.nets import build_model
        net = build_model(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets'), opts)

    sw = Stopwatch()
    sw.wrap(nifti, 'load', 'nifti.load (T1, mask)')
    sw.wrap(nifti, 'save', 'nifti.save (labels, gzip)')
    sw.wrap(base, '_zscore', 'z-score')
    sw.wrap(ndimage, 'binary_dilation', 'ROI dilation x10')
    sw.wrap(base, 'get_mask_voxels', 'ROI voxel list')
    sw.wrap(nifti, 'load_rows', 'prior rows at the ROI voxels (streamed inflate of the atlas)')
    sw.wrap(nifti, 'save_sparse', 'nifti.save_sparse (probabilities, gzip)')
    sw.wrap(net, 'predict_volume', 'GPU path (upload + gather-fused forward + download)')
    if args.post:
        sw.wrap(base, 'post_process_segmentation', 'post_process')

    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        scan = make_subject(d, shape)
        print('synthetic subject %s written in %.1f s' % (shape, time.perf_counter() - t0))
        for r in range(args.repeat):
            sw.t.clear()
            t0 = time.perf_counter()
            base.test_scan(net, scan, opts)
            total = time.perf_counter() - t0
            # nifti.load is nested inside nothing else; everything else is disjoint
            rest = total - sum(sw.t.values())
            print('--- test_scan run %d: %.2f s' % (r, total))
            for k, v in sorted(sw.t.items(), key=lambda kv: -kv[1]):
                print('    %-55s %7.3f s  %5.1f %%' % (k, v, 100 * v / total))
            print('    %-55s %7.3f s  %5.1f %%' % ('other host work (allocation, scatter, casts)', rest, 100 * rest / total))
            sizes = {f: os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith('out_')}
            print('    outputs:', sizes)
        if args.cohort:
            import shutil
            scans = [scan]
            for i in range(1, args.cohort):
                sub = os.path.join(d, 'subj%d' % i)
                shutil.copytree(d, sub, ignore=shutil.ignore_patterns('subj*', 'out_*'))
                scans.append(os.path.join(sub, 'T1.nii.gz'))
            t0 = time.perf_counter()
            for sc in scans:
                base.test_scan(net, sc, opts)
            t_loop = time.perf_counter() - t0
            print('--- cohort of %d subjects: test_scan loop %.2f s (%.2f s/subject)' % (len(scans), t_loop, t_loop / len(scans)))
            for readers in (1, 2, 3, 4):
                t0 = time.perf_counter()
                base.test_cohort(net, scans, opts, readers=readers)
                t_pipe = time.perf_counter() - t0
                print('    test_cohort, %d reader thread(s): %.2f s (%.2f s/subject)' % (readers, t_pipe, t_pipe / len(scans)))


if __name__ == '__main__':
    main()
