"""Net.fit end to end (samples/s per epoch) on a synthetic training set: materialised arrays on the host (streamed in slices),
the same arrays small enough to be resident, and the compact form are timed; the Python loop and the host gathers must
keep up with the GPU step rate (bench.py --mode train)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'sub-cortical_segmentation_amd')):
    sys.path.insert(0, p)
from cnn_cort import synth
from cnn_cort.nets import Net

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
rng = np.random.RandomState(0)
ax, co, sa, at = synth.patch_batch(4096, seed=1)
rep = (n + 4095) // 4096
X = {'in1': np.tile(ax, (rep, 1, 1, 1))[:n], 'in2': np.tile(co, (rep, 1, 1, 1))[:n], 'in3': np.tile(sa, (rep, 1, 1, 1))[:n],
     'in4': np.tile(at, (rep, 1))[:n]}
y = rng.randint(0, 15, n).astype(np.uint8)
for bs in (128, 256):
    for mode, resident in (('streamed from the host', 0), ('resident', 1 << 40)):
        net = Net(max_epochs=2, patience=50, eval_size=0.25, batch_size=bs, verbose=0, device=0, seed=1, fit_resident_bytes=resident)
        t0 = time.time()
        net.fit(X, y)
        durs = [r['dur'] for r in net.train_history_]
        print('batch %d, %s: %d samples, epoch times %s s -> %.0f samples/s (second epoch, training + hold-out pass)'
              % (bs, mode, n, ['%.2f' % d for d in durs], n / durs[-1]), flush=True)

# the compact form: 4 synthetic z-scored volumes, n voxel rows (77 bytes per sample), patches cut inside the kernels
from cnn_cort.base import CompactTrainingSet
vols = [rng.randn(160, 192, 160).astype(np.float32) for _ in range(4)]
labs = [np.zeros(v.shape, np.uint8) for v in vols]
samples = np.stack([rng.randint(0, 4, n), rng.randint(0, 160, n), rng.randint(0, 192, n), rng.randint(0, 160, n)], 1).astype(np.int32)
e = rng.rand(n, 15).astype(np.float32)
for i, lb in enumerate(labs):
    m = samples[:, 0] == i
    lb[samples[m, 1], samples[m, 2], samples[m, 3]] = y[m]            # the device gathers the labels from these volumes
cs = CompactTrainingSet(vols, labs, samples, e / e.sum(1, keepdims=True), y, rng.permutation(n).astype(np.int32))
for bs in (128, 256):
    net = Net(max_epochs=2, patience=50, eval_size=0.25, batch_size=bs, verbose=0, device=0, seed=1)
    net.fit(cs)
    durs = [r['dur'] for r in net.train_history_]
    print('batch %d, compact set (gather-fused): %d samples, epoch times %s s -> %.0f samples/s (second epoch, training + hold-out pass)'
          % (bs, n, ['%.2f' % d for d in durs], n / durs[-1]), flush=True)

# the UNCHANGED call sequence of train_model.py:36-60 on subject folders: load_data -> generate_training_set -> net.fit(dict, y).
# load_data returns lazy patch arrays, generate_training_set lazy views of one compact set, fit() trains gather-fused.
import tempfile
from cnn_cort import base, nifti
from scipy import ndimage
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    for s in range(4):
        folder = os.path.join(d, 'train', 's%d' % s)
        os.makedirs(os.path.join(folder, 'tmp'))
        shape = (160, 192, 160)
        vol = synth.smooth_volume(shape, seed=s, as_int=True)
        atlas = synth.toy_atlas(shape, seed=10 + s)
        lab = synth.toy_labels(shape, atlas)
        # thicken the structures so that the four subjects give ~n samples in all (every label 1..14 voxel + as many boundary voxels)
        fg = ndimage.binary_dilation(lab > 0, iterations=2) & (vol > 0)
        lab = np.where(fg & (lab == 0), 15, lab).astype(np.uint8)
        nifti.save(vol, os.path.join(folder, 'T1.nii.gz'))
        nifti.save(lab, os.path.join(folder, 'gt_15_classes.nii.gz'))
        nifti.save(atlas.astype(np.float32), os.path.join(folder, 'tmp', 'MNI_sub_probabilities.nii.gz'))
    opts = {'train_folder': os.path.join(d, 'train'), 't1_name': 'T1.nii.gz', 'roi_name': 'gt_15_classes.nii.gz', 'patch_size': [32, 32],
            'debug': 'False', 'seed': 5}
    t1 = time.time()
    x_ax, x_co, x_sa, y_ax, x_atl, names = base.load_data(opts)
    tr = base.generate_training_set(x_ax, x_co, x_sa, x_atl, y_ax, opts)
    t2 = time.time()
    N = len(tr[4])
    print('unchanged call sequence: 4 subjects written in %.1f s; load_data + generate_training_set %.2f s for %d samples '
          '(%s)' % (t1 - t0, t2 - t1, N, type(tr[0]).__name__), flush=True)
    for bs in (128, 256):
        net = Net(max_epochs=2, patience=50, eval_size=0.25, batch_size=bs, verbose=0, device=0, seed=1)
        net.fit({'in1': tr[0], 'in2': tr[1], 'in3': tr[2], 'in4': tr[3]}, tr[4])
        durs = [r['dur'] for r in net.train_history_]
        print('batch %d, unchanged call sequence (lazy views -> gather-fused): %d samples, epoch times %s s -> %.0f samples/s (second epoch, '
              'training + hold-out pass)' % (bs, N, ['%.2f' % x for x in durs], N / durs[-1]), flush=True)
