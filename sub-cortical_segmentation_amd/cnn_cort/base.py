"""
Data assembly and the inference driver of the reference, for the HIP net.

Mirrors /root/reference/cnn_cort/base.py (load_data :11-37, load_test_names :41-50, generate_training_set
:53-117, load_patch_vectors :120-184, get_atlas_vectors :187-218, load_patches :221-256, get_patches
:272-308, get_mask_voxels :310-331, load_patch_batch :335-397, test_scan :401-458, post_process_segmentation
:460-480).  Same function names, argument meaning and outputs; what changed is how the work is done:

  * get_patches is one strided-view gather instead of a Python list of per-voxel slices; centre lists are
    int ndarrays [N,3] instead of lists of tuples (len(), slicing and np.stack(centers, axis=1) behave alike).
  * test_scan never materialises patches on the host: the z-scored volume, the ROI centre list and the
    4-D atlas go to the GPU once and libscs cuts the tri-planar patches inside the first conv kernel
    (scs_forward_volume).  Labels and probabilities come from ONE forward pass (the reference runs the net
    twice when out_probabilities is set, base.py:425,435).
  * a training set is kept as volumes + voxel rows + one index permutation (77 bytes per sample instead of 12.3 KB) and
    Net.fit cuts the patches inside the first conv kernel (scs_train_step_volume).  The UNCHANGED call sequence of
    train_model.py:36-60 takes that path: load_data returns lazy per-subject patch arrays (LazyPatches: shape, dtype and
    indexing of the reference's arrays, real arrays as soon as anything asks), generate_training_set recognises them and
    returns the three image arrays as LazyTrainingViews of one CompactTrainingSet, Net.fit recognises those.  Handing either
    function real arrays (or touching the lazy ones with NumPy) gives exactly the reference's materialised arrays;
    load_data_compact / generate_training_set_compact are the explicit form of the same thing.
  * registration (register_masks, base.py:483-551: NiftyReg binaries + atlases that do not ship,
    .MISSING_LARGE_BLOBS) is out of scope: tmp/MNI_sub_probabilities.nii.gz and tmp/MNI_subcortical_mask.nii.gz
    must exist (FileNotFoundError otherwise).

Reference quirks and what this module does about them:
  * predict only under debug == 'True' (base.py:422): FIXED - always predicts.
  * `crop` is a string and always truthy (base.py:367): reproduced for strings ('False' still crops), real
    booleans are honoured.
  * training-path atlas fix-up is dead code (base.py:213-216): reproduced (rows returned as they are).
  * post_process_segmentation's empty-class behaviour (base.py:469-478): reproduced.
  * unseeded negative sampling / permutation: seedable through options['seed'] (default: unseeded, as the
    reference).
"""
import os
import random
import sys
import time

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view
from scipy import ndimage

from . import nifti
from .load_options import is_true

_MODES = ('axial', 'coronal', 'saggital')


# --------------------------------------------------------------------------------------------- names
def load_test_names(options):
    """Subject folders of the inference set -> (T1 paths, folder names) (base.py:41-50)."""
    dir_name = options['test_folder']
    t1_name = options['t1_name']
    subjects = [f for f in sorted(os.listdir(dir_name)) if os.path.isdir(os.path.join(dir_name, f))]
    return [os.path.join(dir_name, s, t1_name) for s in subjects], subjects


def load_only_names(dir_name, mask_name, t1_name, use_t1, size):
    """base.py:259-269."""
    subjects = [f for f in sorted(os.listdir(dir_name)) if os.path.isdir(os.path.join(dir_name, f))]
    return [os.path.join(dir_name, s, t1_name) for s in subjects] if use_t1 else None


# --------------------------------------------------------------------------------------------- patches
def get_patches(image, centers, patch_size=(32, 32), mode='axial'):
    """2-D patches of one view around each centre, zero padded outside the volume (base.py:272-308).
    axial = image[x-h:x+h, y-h:y+h, z], coronal = image[x-h:x+h, y, z-h:z+h], saggital = image[x, y-h:y+h, z-h:z+h]
    (h = patch/2; the centre voxel lands at patch index [h][h]).  Returns ndarray [N, p1, p2] in image's dtype."""
    if mode not in _MODES:
        raise ValueError("mode must be 'axial', 'coronal' or 'saggital'")
    # C order (z fastest) is the layout the window gather below is fast on; volumes straight from nifti.load are in
    # Fortran order, where the same gather is 5-10x slower than the 50 ms this conversion costs
    image = np.ascontiguousarray(image)
    c = np.asarray(centers, dtype=np.int64).reshape(-1, 3)
    p1, p2 = int(patch_size[0]), int(patch_size[1])
    if c.shape[0] == 0:
        return np.zeros((0, p1, p2), dtype=image.dtype)
    ax = {'axial': (0, 1), 'coronal': (0, 2), 'saggital': (1, 2)}[mode]
    h1, h2 = p1 // 2, p2 // 2
    pad = [(0, 0)] * 3
    pad[ax[0]] = (h1, p1 - h1)
    pad[ax[1]] = (h2, p2 - h2)
    padded = np.pad(image, pad, mode='constant', constant_values=0)
    win = sliding_window_view(padded, (p1, p2), axis=ax)         # [..., p1, p2]; window origin = padded index
    # after padding, the window that starts at the unpadded centre coordinate is the centred patch
    return np.ascontiguousarray(win[c[:, 0], c[:, 1], c[:, 2]])


def get_mask_voxels(mask, size=None, rng=None):
    """Coordinates of the non-zero voxels in C order as int64 [N,3] (base.py:310-331).  With `size` the list is
    shuffled (stdlib `random`, or the random.Random given as `rng`) and truncated, as the reference does."""
    indices = np.stack(np.nonzero(mask), axis=1).astype(np.int64)
    if size is not None:
        perm = list(range(indices.shape[0]))
        (rng or random).shuffle(perm)
        indices = indices[np.asarray(perm[:size], dtype=np.int64)]
    return indices


def _zscore(image, as_float32):
    """(image - mean) / std over the non-zero voxels, population std (base.py:146 training, :358 inference)."""
    nz = image[np.nonzero(image)]
    if as_float32:
        return (image.astype(np.float32) - nz.mean()) / nz.std()
    return (image - nz.mean()) / nz.std()


# --------------------------------------------------------------------------------------------- lazy patch arrays
class LazyPatches:
    """What load_patch_vectors returns per subject and view instead of an [n, p, p] array: the (volume, centres, view) that
    DEFINE that array.  It has the array's shape / dtype / len, indexes like it (an int gives one [p, p] patch, a slice or an
    index array the patches of those rows, anything else - tuples, masks - goes through the materialised array), and becomes the
    real array the moment anything asks for one (np.asarray, np.concatenate, arithmetic: get_patches on demand) - so code written
    against the reference's lists of arrays keeps working unchanged, while generate_training_set and Net.fit, which know the
    type, never build the 12.3 KB-per-sample patches at all (they are cut inside the first GPU kernels)."""

    __array_priority__ = 100.0

    def __init__(self, volume, centers, patch_size, mode):
        self.volume = volume
        self.centers = np.asarray(centers, dtype=np.int64).reshape(-1, 3)
        self.patch_size = (int(patch_size[0]), int(patch_size[1]))
        self.mode = mode

    shape = property(lambda self: (self.centers.shape[0],) + self.patch_size)
    dtype = property(lambda self: self.volume.dtype)
    ndim = 3
    size = property(lambda self: self.centers.shape[0] * self.patch_size[0] * self.patch_size[1])

    def __len__(self):
        return self.centers.shape[0]

    def rows(self, idx):
        return get_patches(self.volume, self.centers[idx].reshape(-1, 3), self.patch_size, self.mode)

    def __array__(self, dtype=None, copy=None):
        a = get_patches(self.volume, self.centers, self.patch_size, self.mode)
        return a if dtype is None else a.astype(dtype, copy=False)

    def astype(self, dtype, **kw):
        return np.asarray(self).astype(dtype, **kw)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.rows(np.asarray([key]))[0]
        if isinstance(key, slice) or (isinstance(key, np.ndarray) and key.ndim == 1 and key.dtype.kind in 'iu') or \
                (isinstance(key, list) and all(isinstance(k, (int, np.integer)) for k in key)):
            return self.rows(np.arange(len(self))[key] if isinstance(key, slice) else np.asarray(key))
        return np.asarray(self)[key]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class LazyTrainingView:
    """One of the three [N, 1, p, p] float32 image arrays generate_training_set returns, unmaterialised: a view name + the
    CompactTrainingSet that defines all three.  Same contract as LazyPatches (shape, dtype, indexing, np.asarray); Net.fit takes
    the gather-fused path when it is handed the three views of one set."""

    __array_priority__ = 100.0

    def __init__(self, compact, view):
        self.compact = compact
        self.view = int(view)                     # 0 axial, 1 coronal, 2 saggital

    shape = property(lambda self: (len(self.compact), 1, 32, 32))
    dtype = np.dtype(np.float32)
    ndim = 4
    size = property(lambda self: len(self.compact) * 1024)

    def __len__(self):
        return len(self.compact)

    def rows(self, rows):
        cs = self.compact
        smp = cs.samples[cs.index[rows]]
        x = np.zeros((smp.shape[0], 32, 32), np.float32)
        for v, vol in enumerate(cs.volumes):
            sel = np.nonzero(smp[:, 0] == v)[0]
            if sel.size:
                x[sel] = get_patches(vol, smp[sel, 1:4], (32, 32), _MODES[self.view])
        return x[:, None]

    def __array__(self, dtype=None, copy=None):
        a = self.rows(np.arange(len(self)))
        return a if dtype is None else a.astype(dtype, copy=False)

    def astype(self, dtype, **kw):
        return np.asarray(self).astype(dtype, **kw)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.rows(np.asarray([key]))[0]
        if isinstance(key, slice) or (isinstance(key, np.ndarray) and key.ndim == 1 and key.dtype.kind in 'iu'):
            return self.rows(np.arange(len(self))[key] if isinstance(key, slice) else key)
        return np.asarray(self)[key]


def compact_from_views(X, y):
    """The CompactTrainingSet behind a fit() input dict whose 'in1'..'in3' are the three LazyTrainingViews of ONE set, carrying the
    prior rows and labels the caller actually passed ('in4', y - in training order, possibly edited since
    generate_training_set returned them); None when the dict is anything else."""
    v = [X.get(k) for k in ('in1', 'in2', 'in3')]
    if not all(isinstance(a, LazyTrainingView) for a in v):
        return None
    cs = v[0].compact
    if not (v[1].compact is cs and v[2].compact is cs and [a.view for a in v] == [0, 1, 2]):
        return None
    n = len(cs)
    if cs.samples.shape[0] != n:
        return None
    atlas = np.asarray(X['in4'], dtype=np.float32)
    yy = np.asarray(y)
    if atlas.shape != (n, 15) or yy.shape != (n,):
        return None
    a_s = np.empty_like(atlas)
    y_s = np.empty((n,), np.uint8)
    a_s[cs.index] = atlas                          # training order -> sample order
    if yy.size and (yy.min() < 0 or yy.max() > 255):
        raise ValueError('labels must lie in 0..14')
    y_s[cs.index] = yy.astype(np.uint8)
    return CompactTrainingSet(cs.volumes, cs.labels, cs.samples, a_s, y_s, cs.index)


# --------------------------------------------------------------------------------------------- training data
def _subjects(dir_name):
    return [f for f in sorted(os.listdir(dir_name)) if os.path.isdir(os.path.join(dir_name, f))]


def load_patch_vectors(name, label_name, dir_name, size, random_state=42, balance_neg=True, rng=None):
    """Per-subject training patches of all three views (base.py:120-184): every voxel with label 1..14, then an
    equal number of randomly chosen label-15 (boundary) voxels; image AND label patches."""
    subjects = _subjects(dir_name)
    image_names = [os.path.join(dir_name, s, name) for s in subjects]
    images = [nifti.load(n).get_data() for n in image_names]
    images_norm = [_zscore(im, True) for im in images]
    labels = [nifti.load(os.path.join(dir_name, s, label_name)).get_data() for s in subjects]

    pos = [get_mask_voxels(np.logical_and(m > 0, m < 15)) for m in labels]
    if balance_neg:
        neg = [get_mask_voxels(m == 15, size=len(p), rng=rng) for m, p in zip(labels, pos)]
    else:
        neg = [get_mask_voxels(m == 15) for m in labels]
    vox_positions = [np.concatenate([p, n]) for p, n in zip(pos, neg)]
    # the patch arrays stay implicit (LazyPatches: positives then negatives, as the reference concatenates them); C-order copies
    # of the volumes, which is what get_patches - and the GPU gather - read
    images_norm = [np.ascontiguousarray(im, dtype=np.float32) for im in images_norm]
    labels = [np.ascontiguousarray(lb) for lb in labels]
    out = {}
    for mode in _MODES:
        out['x_' + mode] = [LazyPatches(im, c, size, mode) for im, c in zip(images_norm, vox_positions)]
        out['y_' + mode] = [LazyPatches(lb, c, size, mode) for lb, c in zip(labels, vox_positions)]
    return (out['x_axial'], out['y_axial'], out['x_coronal'], out['y_coronal'], out['x_saggital'], out['y_saggital'],
            vox_positions, image_names)


def get_atlas_vectors(dir_name, centers, t1_names):
    """15-vector of the registered atlas at each patch centre (base.py:187-218)."""
    subjects = _subjects(dir_name)
    vectors = []
    for subject, c in zip(subjects, centers):
        atlas_path = os.path.join(dir_name, subject, 'tmp', 'MNI_sub_probabilities.nii.gz')
        if not os.path.exists(atlas_path):
            raise FileNotFoundError('%s is missing: run the NiftyReg registration of the reference first '
                                    '(register_masks, base.py:483-551, is out of scope here)' % atlas_path)
        a = nifti.load(atlas_path).get_data()
        c = np.asarray(c, dtype=np.int64)
        vectors.append(a[c[:, 0], c[:, 1], c[:, 2]])
    return vectors


def load_patches(dir_name, mask_name, t1_name, size, seeds=None, balance_neg=True, rng=None):
    """base.py:221-256."""
    print('    --> Loading ' + t1_name + ' images')
    x_axial, y_axial, x_cor, y_cor, x_sag, y_sag, centers, t1_names = load_patch_vectors(
        t1_name, mask_name, dir_name, size, 0, balance_neg, rng)
    x_atlas = get_atlas_vectors(dir_name, centers, t1_names)
    return x_axial, y_axial, x_cor, y_cor, x_sag, y_sag, x_atlas, t1_names


def load_data(options):
    """Patches of every training subject, all views (base.py:11-37).
    Returns (x_axial, x_cor, x_sag, y_axial, x_atlas, names): lists indexed by subject."""
    rng = random.Random(options['seed']) if options.get('seed') is not None else None
    (x_axial, y_axial, x_cor, y_cor, x_sag, y_sag, x_atlas, names) = load_patches(
        dir_name=options['train_folder'], t1_name=options['t1_name'], mask_name=options['roi_name'],
        size=tuple(options['patch_size']), rng=rng)
    return x_axial, x_cor, x_sag, y_axial, x_atlas, names


def generate_training_set(x_axial, x_coronal, x_saggital, x_atlas, y, options, randomize=True):
    """Concatenate subjects, take the centre label, map 15 -> 0, apply ONE permutation to all five arrays, add
    the channel axis (base.py:53-117).  Returns (x_axial [N,1,p,p], x_cor, x_sag, x_atlas [N,15], y [N] uint8)."""
    lazy = _lazy_training_set(x_axial, x_coronal, x_saggital, x_atlas, y, options, randomize)
    if lazy is not None:
        return lazy
    x_train_axial = np.concatenate(x_axial, axis=0).astype('float32')
    x_train_cor = np.concatenate(x_coronal, axis=0).astype('float32')
    x_train_sag = np.concatenate(x_saggital, axis=0).astype('float32')
    x_train_atlas = np.concatenate(x_atlas, axis=0).astype('float32')
    y_train = np.concatenate(y, axis=0).astype('uint8')
    y_train = y_train[:, y_train.shape[1] // 2, y_train.shape[2] // 2]
    y_train = np.squeeze(y_train)
    y_train[y_train == 15] = 0
    if randomize:
        seed = options.get('seed')
        if seed is None:
            seed = np.random.randint(np.iinfo(np.int32).max)
        # the reference re-seeds before each of the five permutations, i.e. applies one permutation five times
        np.random.seed(seed)
        perm = np.random.permutation(x_train_axial.shape[0])
        x_train_axial, x_train_cor, x_train_sag = x_train_axial[perm], x_train_cor[perm], x_train_sag[perm]
        y_train, x_train_atlas = y_train[perm], x_train_atlas[perm]
    x_train_axial = np.expand_dims(x_train_axial, axis=1)
    x_train_cor = np.expand_dims(x_train_cor, axis=1)
    x_train_sag = np.expand_dims(x_train_sag, axis=1)
    if is_true(options.get('debug')):
        print("    --> X_TRAIN: ", x_train_axial.shape[0], x_train_axial.shape)
        print("    --> Y_TRAIN POS: ", y_train[y_train > 0].shape[0])
        print("    --> Y_TRAIN NEG: ", y_train[y_train == 0].shape[0])
    return x_train_axial, x_train_cor, x_train_sag, x_train_atlas, y_train


def _lazy_training_set(x_axial, x_coronal, x_saggital, x_atlas, y, options, randomize):
    """generate_training_set on the lists load_data returns (LazyPatches): the same five outputs - the three image arrays as
    LazyTrainingViews of one CompactTrainingSet (same permutation as the materialising code below under the same seed), prior
    rows and labels as real arrays.  None when any input is not what load_data returned (a real array, a different patch size,
    views that disagree on their voxels): the materialising code then runs, on np.asarray of whatever it was given."""
    lists = (x_axial, x_coronal, x_saggital, y)
    if not all(isinstance(l, (list, tuple)) and len(l) == len(x_axial) and len(l) > 0 for l in lists + (x_atlas,)):
        return None
    if not all(isinstance(a, LazyPatches) and a.patch_size == (32, 32) for l in lists for a in l):
        return None
    for i in range(len(x_axial)):
        ax, co, sa, lb = x_axial[i], x_coronal[i], x_saggital[i], y[i]
        if (ax.mode, co.mode, sa.mode) != _MODES or not (co.volume is ax.volume and sa.volume is ax.volume):
            return None
        if not (co.centers is ax.centers or np.array_equal(co.centers, ax.centers)) or not (sa.centers is ax.centers or np.array_equal(sa.centers, ax.centers)):
            return None
        if lb.volume.shape != ax.volume.shape or not (lb.centers is ax.centers or np.array_equal(lb.centers, ax.centers)):
            return None
        if ax.volume.dtype != np.float32 or np.asarray(x_atlas[i]).shape != (len(ax), 15):
            return None
    data = CompactTrainingData([a.volume for a in x_axial], [np.ascontiguousarray(l.volume).astype(np.uint8) for l in y],
                               [a.centers for a in x_axial], [np.asarray(a) for a in x_atlas], None)
    cs = generate_training_set_compact(data, options, randomize=randomize)
    return (LazyTrainingView(cs, 0), LazyTrainingView(cs, 1), LazyTrainingView(cs, 2), cs.atlas[cs.index], cs.y[cs.index])


# --------------------------------------------------------------------------------------------- compact training set
class CompactTrainingData:
    """What load_data returns, without the patches: per subject the z-scored volume, the label volume, the sampled voxel
    positions (positives then negatives, as load_patch_vectors orders them) and the prior vectors at those voxels."""

    def __init__(self, volumes, labels, positions, atlas, names):
        self.volumes, self.labels, self.positions, self.atlas, self.names = volumes, labels, positions, atlas, names


class CompactTrainingSet:
    """A training set as the GPU consumes it (scs_train_step_volume): volumes + one (volume, x, y, z) row, one prior row and
    one label per sample + the permutation as an index array - 77 bytes per sample instead of 12.3 KB of patches.
    `materialize()` gives back exactly what generate_training_set returns for the same data and seed."""

    def __init__(self, volumes, labels, samples, atlas, y, index):
        self.volumes = [np.ascontiguousarray(v, dtype=np.float32) for v in volumes]
        self.labels = [np.ascontiguousarray(v, dtype=np.uint8) for v in labels]
        self.samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 4)
        self.atlas = np.ascontiguousarray(atlas, dtype=np.float32).reshape(-1, 15)
        self.y = np.ascontiguousarray(y, dtype=np.uint8).reshape(-1)
        self.index = np.ascontiguousarray(index, dtype=np.int32).reshape(-1)
        n = self.samples.shape[0]
        if not (self.atlas.shape[0] == n and self.y.shape[0] == n and self.index.shape[0] <= n):
            raise ValueError('samples, atlas rows, labels and index disagree on the number of samples')
        # the device reads the label volumes with the image volumes' geometry (scs_gather_labels): a mismatch would be an
        # out-of-bounds read there, so it is refused here
        if len(self.labels) != len(self.volumes) or any(v.ndim != 3 or lb.shape != v.shape for v, lb in zip(self.volumes, self.labels)):
            raise ValueError('every label volume must have the shape of its image volume (3-D)')
        if n and not (0 <= int(self.samples[:, 0].min()) and int(self.samples[:, 0].max()) < len(self.volumes)):
            raise ValueError('sample rows name a volume outside the set')
        if self.index.size and not (0 <= int(self.index.min()) and int(self.index.max()) < n):
            raise ValueError('index entries must be rows of `samples`')

    def __len__(self):
        return int(self.index.shape[0])

    def descriptors(self):
        """(offsets into the concatenated volume buffer, dims) per volume - the scs_volume table (include/scs.h)."""
        offs, dims, at = [], [], 0
        for v in self.volumes:
            offs.append(at)
            dims.append(v.shape)
            at += v.size
        return offs, dims

    def materialize(self, lo=0, hi=None):
        """Host patches of rows index[lo:hi]: (x_axial [n,1,32,32], x_cor, x_sag, x_atlas [n,15], y [n]) - what
        generate_training_set returns (base.py:53-117)."""
        rows = self.index[lo:hi]
        smp = self.samples[rows]
        out = []
        for mode in _MODES:
            x = np.zeros((rows.shape[0], 32, 32), np.float32)
            for v, vol in enumerate(self.volumes):
                sel = np.nonzero(smp[:, 0] == v)[0]
                if sel.size:
                    x[sel] = get_patches(vol, smp[sel, 1:4], (32, 32), mode)
            out.append(x[:, None])
        return out[0], out[1], out[2], self.atlas[rows], self.y[rows]


def load_data_compact(options):
    """load_data (base.py:11-37) without cutting patches: same subjects, same voxel sampling (every voxel with label 1..14,
    then as many randomly chosen label-15 voxels, same generator state as load_data under options['seed']), same prior
    vectors - but the tri-planar patches stay implicit in (volume, voxel) and are cut on the GPU."""
    rng = random.Random(options['seed']) if options.get('seed') is not None else None
    dir_name, name, label_name = options['train_folder'], options['t1_name'], options['roi_name']
    subjects = _subjects(dir_name)
    image_names = [os.path.join(dir_name, s, name) for s in subjects]
    volumes = [np.ascontiguousarray(_zscore(nifti.load(n).get_data(), True), dtype=np.float32) for n in image_names]
    labels = [np.ascontiguousarray(nifti.load(os.path.join(dir_name, s, label_name)).get_data()) for s in subjects]
    pos = [get_mask_voxels(np.logical_and(m > 0, m < 15)) for m in labels]
    neg = [get_mask_voxels(m == 15, size=len(p), rng=rng) for m, p in zip(labels, pos)]
    positions = [np.concatenate([p, n]) for p, n in zip(pos, neg)]
    atlas = get_atlas_vectors(dir_name, positions, image_names)
    return CompactTrainingData(volumes, [lb.astype(np.uint8) for lb in labels], positions, atlas, image_names)


def generate_training_set_compact(data, options, randomize=True):
    """generate_training_set (base.py:53-117) on the compact form: subjects concatenated, centre label with 15 -> 0, and the
    ONE permutation the reference applies to all five arrays kept as an index array (same seed -> same order)."""
    samples = np.concatenate([np.concatenate([np.full((p.shape[0], 1), v, np.int64), p], axis=1) for v, p in enumerate(data.positions)])
    atlas = np.concatenate(data.atlas, axis=0).astype('float32')
    y = np.concatenate([lb[p[:, 0], p[:, 1], p[:, 2]] for lb, p in zip(data.labels, data.positions)]).astype('uint8')
    y[y == 15] = 0
    n = samples.shape[0]
    index = np.arange(n)
    if randomize:
        seed = options.get('seed')
        if seed is None:
            seed = np.random.randint(np.iinfo(np.int32).max)
        np.random.seed(seed)
        index = np.random.permutation(n)
    if is_true(options.get('debug')):
        print("    --> X_TRAIN: ", n, (n, 1, 32, 32))
        print("    --> Y_TRAIN POS: ", y[y > 0].shape[0])
        print("    --> Y_TRAIN NEG: ", y[y == 0].shape[0])
    return CompactTrainingSet(data.volumes, data.labels, samples, atlas, y, index)


# --------------------------------------------------------------------------------------------- inference
def _crop_enabled(options):
    v = options.get('crop', 'True')
    return bool(v)          # the reference tests the config STRING (base.py:367): any non-empty string crops


def _scan_inputs(scan_name, options, dilate=None, net=None):
    """Everything test_scan / load_patch_batch need from disk: normalised volume, ROI centres, and the prior rows
    of those centres.  The 15-channel prior volume itself (~0.8 GB for a 256x256x192 subject) is never held:
    nifti.load_rows streams the file once and keeps the ROI entries.
    With `net` (a cnn_cort.nets.Net) the whole-volume work happens on the GPU: the raw T1 and the raw mask go up as they lie
    in their files, the z-score (net.normalize_volume) and the dilated ROI with its voxel list (net.roi_centers) are computed
    there, and the normalised volume / centre list STAY there for predict_volume (image_norm is then a device tensor and a
    sixth entry holds the device copy of the centres).  `dilate(mask, iterations)` alone replaces scipy's binary_dilation."""
    dir_name = os.path.dirname(scan_name)
    t1 = nifti.load(scan_name)
    image = t1.get_data()
    atlas_name = os.path.join(dir_name, 'tmp', 'MNI_sub_probabilities.nii.gz')
    mask_name = os.path.join(dir_name, 'tmp', 'MNI_subcortical_mask.nii.gz')
    if not os.path.exists(atlas_name):
        raise FileNotFoundError('%s is missing: run the NiftyReg registration of the reference first '
                                '(register_masks, base.py:483-551, is out of scope here)' % atlas_name)
    on_device = net is not None and hasattr(net, 'normalize_volume') and hasattr(net, 'roi_centers')
    centers_dev = None
    mask_atlas = nifti.load(mask_name).get_data() if _crop_enabled(options) else None      # (post-processing reads it again: kept)
    if on_device:
        image_norm = net.normalize_volume(image)
        if mask_atlas is not None:
            centers, centers_dev = net.roi_centers(mask_atlas, 10)
        else:
            centers, centers_dev = net.roi_centers(image, 0)
    else:
        image_norm = _zscore(image, False)
        if mask_atlas is not None:
            b_mask = dilate(mask_atlas, 10) if dilate else ndimage.binary_dilation(mask_atlas, iterations=10)
            centers = get_mask_voxels(b_mask.astype(bool))
        else:
            centers = get_mask_voxels(image.astype(bool))
    if is_true(options.get('debug')):
        print("    -->  num of samples to test:", len(centers))
    return t1, image, image_norm, centers, atlas_rows_from_file(atlas_name, centers, image.shape), centers_dev, mask_atlas


def _background_rule(v):
    """Rows without any prior become background: channel 14 = 1 (base.py:391-394)."""
    v = np.asarray(v).astype(np.float32)
    v[v.sum(axis=1) == 0, 14] = 1
    return v


def atlas_vectors_at(atlas_image, centers):
    """[N,15] float32 prior rows with the inference-path background rule (base.py:388-394)."""
    c = np.asarray(centers, dtype=np.int64).reshape(-1, 3)
    return _background_rule(atlas_image[c[:, 0], c[:, 1], c[:, 2]])


def atlas_rows_from_file(atlas_name, centers, volume_shape=None):
    """atlas_vectors_at(load(atlas_name).get_data(), centers) without loading the volume."""
    rows = nifti.load_rows(atlas_name, centers)
    if rows.ndim != 2 or rows.shape[1] != 15:
        raise ValueError('%s: expected a 4-D prior with 15 channels' % atlas_name)
    return _background_rule(rows)


def load_patch_batch(scan_name, options, datatype=np.float32):
    """Generator over host-side test batches, as the reference yields them (base.py:335-397):
    (axial [B,1,p,p], coronal, saggital, atlas [B,15], centers).  Kept for API compatibility and for tests;
    test_scan itself uses the gather-fused GPU path instead."""
    _, _, image_norm, centers_all, atlas_rows, _, _ = _scan_inputs(scan_name, options)
    bs = options['test_batch_size']
    for i in range(0, len(centers_all), bs):
        centers = centers_all[i:i + bs]
        views = [np.stack([get_patches(image_norm, centers, options['patch_size'], mode=m).astype(datatype)], axis=1)
                 for m in _MODES]
        yield views[0], views[1], views[2], atlas_rows[i:i + bs], centers


def _scan_predict(net, inputs, options):
    t1, raw, image_norm, centers, atlas_rows, centers_dev, _ = inputs
    want_probs = is_true(options.get('out_probabilities'))
    # options['towers'] ('auto' | 'patch' | 'shared', nets.Net.predict_volume) is handed on only when set: other nets need not know it
    how = {'towers': options['towers']} if 'towers' in options else {}
    # the priors stay on the host; only the ROI rows travel (base.py:388-394 rule already applied)
    if centers_dev is not None:                     # volume and voxel list are already on the device (_scan_inputs with a net)
        return net.predict_volume(image_norm, centers_dev, atlas_rows=atlas_rows, want_probs=want_probs, batch=options['test_batch_size'], **how)
    return net.predict_volume(np.ascontiguousarray(image_norm, dtype=np.float32), centers, atlas_rows=atlas_rows,
                              want_probs=want_probs, batch=options['test_batch_size'], **how)


def _scan_outputs(image_path, inputs, labels, probs, options):
    t1, raw, _, centers, _, _, mask_atlas = inputs
    image = np.zeros_like(raw)
    x, y, z = centers[:, 0], centers[:, 1], centers[:, 2]
    image[x, y, z] = labels
    if probs is not None:
        # the reference fills a dense float64 [X,Y,Z,15] array (base.py:417,438) and saves it; the same file is
        # written here from the ROI rows alone (1.5 GB dense for a 256x256x192 subject, ~99 % zeros)
        nifti.save_sparse(os.path.join(image_path, 'out_subcortical_prob.nii.gz'), tuple(raw.shape) + (15,),
                          np.float64, centers, probs, like=t1)
    if is_true(options.get('post_process')):
        box = tuple(slice(int(lo), int(hi) + 1) for lo, hi in zip(centers.min(0), centers.max(0))) if len(centers) else None
        seg = post_process_mask(image, mask_atlas, box) if mask_atlas is not None else post_process_segmentation(image_path, image, box)
        nifti.save_mostly_constant(seg, os.path.join(image_path, 'out_subcortical_seg_prec.nii.gz'), like=t1)
    else:
        nifti.save_mostly_constant(image, os.path.join(image_path, 'out_subcortical_rawseg.nii.gz'), like=t1)


def test_scan(net, test_scan, options):
    """Segment one T1 volume; writes out_subcortical_{prob,seg_prec,rawseg}.nii.gz next to it and returns the
    elapsed minutes (base.py:401-458)."""
    s_time = time.time()
    inputs = _scan_inputs(test_scan, options, dilate=getattr(net, 'dilate_mask', None), net=net)
    labels, probs = _scan_predict(net, inputs, options)
    _scan_outputs(os.path.dirname(test_scan), inputs, labels, probs, options)
    return (time.time() - s_time) / 60.0


def test_cohort(net, scan_paths, options, shard=True, readers=4, writers=2):
    """test_scan over a list of T1 paths (the loop of train_model.py:74-78) as a three-stage pipeline: while the
    GPU segments subject i, `readers` helper threads already read the next subjects (gunzip of the prior is the
    longest stage) and `writers` others write the files of the subjects before it (with the towers shared between voxels the
    GPU needs 0.02 s per subject and the host threads set the rate).  Same files as calling test_scan per path.  Under torchrun every
    rank takes every world_size-th subject (no collective).  Returns [(path, minutes)] for this rank's subjects,
    minutes = that subject's own load + predict + write time."""
    from concurrent.futures import ThreadPoolExecutor
    from . import dist
    paths = dist.shard_list(list(scan_paths)) if shard else list(scan_paths)
    out = []
    if not paths:
        return out

    def timed(fn, *a):
        t0 = time.time()
        r = fn(*a)
        return r, time.time() - t0

    on_device = hasattr(net, 'normalize_volume') and hasattr(net, 'roi_centers') and hasattr(net, 'side_stream')

    def read(p):
        """Reader thread: files -> inputs.  With a GPU net the z-score and the ROI voxel list are computed on the device from
        the raw volumes, on a stream of this thread's own (complete when the call returns: net.roi_centers waits for it)."""
        if on_device:
            with net.side_stream():
                return _scan_inputs(p, options, net=net)
        return _scan_inputs(p, options, dilate=getattr(net, 'dilate_mask', None))

    ahead = max(1, int(readers))                   # subjects being read while one is on the GPU
    with ThreadPoolExecutor(ahead) as loader, ThreadPoolExecutor(max(1, int(writers))) as writer:   # files of different subjects are independent
        pending = [loader.submit(timed, read, p) for p in paths[:ahead]]
        writes, gpu_times = [], []
        for i, p in enumerate(paths):
            inputs, t_load = pending.pop(0).result()
            if i + ahead < len(paths):
                pending.append(loader.submit(timed, read, paths[i + ahead]))
            (labels, probs), t_gpu = timed(_scan_predict, net, inputs, options)
            gpu_times.append(t_gpu)
            writes.append((p, t_load + t_gpu, writer.submit(timed, _scan_outputs, os.path.dirname(p), inputs, labels, probs, options)))
        stage = [0.0, 0.0]
        for p, t, fut in writes:
            _, t_write = fut.result()
            out.append((p, (t + t_write) / 60.0))
            stage[0] += t; stage[1] += t_write
    if os.environ.get('SCS_COHORT_TIMES'):          # diagnostic: mean seconds a subject spends in each stage (not wall: stages overlap)
        sys.stderr.write('test_cohort: %d subjects, load + predict %.3f s, of which predict %.3f s (caller thread), write %.3f s per subject\n'
                         % (len(out), stage[0] / len(out), sum(gpu_times) / len(out), stage[1] / len(out)))
    return out


def post_process_mask(input_mask, atlas, roi_box=None):
    """Array-level body of post_process_segmentation (base.py:464-480): per class 1..14 keep the connected
    component with the largest overlap with the atlas mask.  np.unique(labels) includes the background
    component 0 and the argmax INDEX is used as the component label, exactly as the reference does (so a class
    without any atlas overlap - or without any voxel - paints its complement).

    Same result as the literal procedure (oracle/patches_ref.py), at a fraction of its passes over the volume (24 s -> 0.02 s
    on a 256x256x192 subject): each class is labelled inside its own bounding box (component numbering follows the raster
    order of first voxels, which a crop preserves); the boxes are looked for inside `roi_box` only (a tuple of slices outside
    which input_mask is known to be 0 - test_scan passes the bounding box of the voxels it classified); and the whole-volume
    paints of the quirk are not executed one by one: a class without voxels sets the fill value and forgets everything painted
    before it, which is all the reference's `filtered_mask[...] = l` amounts to."""
    input_mask = np.asarray(input_mask)
    atlas = np.asarray(atlas)
    if roi_box is None:
        roi_box = tuple(slice(0, d) for d in input_mask.shape)
    sub = input_mask[roi_box]
    origin = [sl.start for sl in roi_box]
    boxes = ndimage.find_objects(np.where((sub >= 1) & (sub <= 14), sub, 0).astype(np.int32), max_label=14)
    fill = 0                                                  # value of every voxel no later paint touched
    paints = []                                               # ('box', box, component mask, l) / ('not', l) in order
    for l in range(1, 15):
        box = boxes[l - 1]
        if box is None:
            # no voxel of this class: the only "component" is the background, index 0 -> everything becomes l
            fill, paints = l, []
            continue
        box = tuple(slice(b.start + o, b.stop + o) for b, o in zip(box, origin))
        th = input_mask[box] == l
        labels, num = ndimage.label(th)
        counts = np.bincount(labels.ravel(), weights=(th & (atlas[box] != 0)).ravel().astype(np.float64), minlength=num + 1)
        has_background = th.size < input_mask.size or not th.all()
        if has_background:
            pick = int(np.argmax(counts))                       # label_list = [0, 1, .., num]: index == component id
            if pick == 0:
                paints.append(('not', l))                       # the background component of this class: everything but its voxels
            else:
                paints.append(('box', box, labels == pick, l))
        else:
            # the whole volume is class l: label_list = [1], argmax index 0 selects component "0", which is empty
            pass
    filtered_mask = np.full(input_mask.shape, fill, dtype=input_mask.dtype)
    for p in paints:
        if p[0] == 'not':
            filtered_mask[input_mask != p[1]] = p[1]
        else:
            filtered_mask[p[1]][p[2]] = p[3]
    return filtered_mask


def post_process_segmentation(image_folder, input_mask, roi_box=None):
    """base.py:460-480."""
    atlas = nifti.load(os.path.join(image_folder, 'tmp', 'MNI_subcortical_mask.nii.gz')).get_data()
    return post_process_mask(input_mask, atlas, roi_box)
