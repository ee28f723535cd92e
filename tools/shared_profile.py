"""One synthetic BASELINE-configs[4] subject through Net.predict_volume with the towers shared between voxels (and per voxel with
--patch), a few times - the workload to put under `rocprofv3 --kernel-trace --stats`.   python tools/shared_profile.py [--patch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
sys.path.insert(0, ROOT)

import numpy as np
import torch
from scipy import ndimage

from cnn_cort import synth
from cnn_cort.nets import build_model


def main():
    shape = (256, 256, 192)
    core = synth.ellipsoid_mask(shape, radii=(0.3, 0.27, 0.33))
    mask = ndimage.binary_erosion(core, iterations=40) if False else core
    # the same ROI construction as bench.make_cohort: a registered sub-cortical mask dilated 10 times
    import bench
    import tempfile
    root = tempfile.mkdtemp(prefix='scs_shared_')
    scan = bench.make_cohort(root, 1, seed=0)[0]
    from cnn_cort import base, nifti
    raw = nifti.load(scan).get_data()
    m = nifti.load(os.path.join(os.path.dirname(scan), 'tmp', 'MNI_subcortical_mask.nii.gz')).get_data()
    roi = ndimage.binary_dilation(m, iterations=10)
    centers = np.ascontiguousarray(np.argwhere(roi), dtype=np.int32)
    opts = {'experiment': 'miccai2012_v1', 'net_verbose': 0, 'train_split': 0.25, 'max_epochs': 1, 'patience': 20, 'patch_size': [32, 32],
            'load_weights': 'True', 'batch_size': 128, 'device': 0}
    net = build_model(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets'), opts)
    vol = net.normalize_volume(raw)
    atlas = np.random.RandomState(0).dirichlet(np.ones(15), len(centers)).astype(np.float32)
    d_cen = torch.from_numpy(centers).cuda()
    how = 'patch' if '--patch' in sys.argv else 'shared'
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.predict_volume(vol, d_cen, atlas_rows=atlas, want_probs=True, towers=how)
        torch.cuda.synchronize()
        print('%s towers, %d voxels in a box of %d: %.4f s' % (how, len(centers), np.prod(centers.max(0) - centers.min(0) + 1), time.perf_counter() - t0))
    import shutil
    shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
