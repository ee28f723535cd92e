"""
TEST INFRASTRUCTURE - NOT PRODUCT CODE (rules in oracle/np_oracle.py).

Literal Python-3 restatement of the reference's patch-extraction / batching helpers, kept deliberately
close to the reference's control flow (per-voxel slice objects, Python lists of tuples) so that the
vectorised / GPU product code in sub-cortical_segmentation_amd/cnn_cort/base.py can be checked against it
index for index.  These functions operate on in-memory arrays; file IO is not part of the oracle.

PARITY UNPINNED by reference fixtures (the reference has none); what is restated here is plain NumPy
indexing whose behaviour is defined by the reference source itself:
  get_patches            /root/reference/cnn_cort/base.py:272-308
  get_mask_voxels        /root/reference/cnn_cort/base.py:310-331
  zscore (inference)     /root/reference/cnn_cort/base.py:357-358
  zscore (training)      /root/reference/cnn_cort/base.py:146
  atlas vectors + fix    /root/reference/cnn_cort/base.py:388-394
  load_patch_batch       /root/reference/cnn_cort/base.py:380-397
  training sampler       /root/reference/cnn_cort/base.py:153-181
  generate_training_set  /root/reference/cnn_cort/base.py:78-110
  scatter in test_scan   /root/reference/cnn_cort/base.py:430-440
  post_process           /root/reference/cnn_cort/base.py:460-480
"""
import random

import numpy as np
from scipy import ndimage


def get_patches(image, centers, patch_size=(32, 32), mode='axial'):
    """base.py:272-308. Python-2 integer '/' restated as '//'. Returns a list of [p1,p2] arrays."""
    if mode == 'axial':
        patch_size = (patch_size[0], patch_size[1], 1)
    if mode == 'coronal':
        patch_size = (patch_size[0], 1, patch_size[1])
    if mode == 'saggital':
        patch_size = (1, patch_size[0], patch_size[1])
    patch_half = tuple([idx // 2 for idx in patch_size])
    new_centers = [[c + h for c, h in zip(center, patch_half)] for center in centers]
    padding = tuple((idx, size - idx) for idx, size in zip(patch_half, patch_size))
    new_image = np.pad(image, padding, mode='constant', constant_values=0)
    slices = [tuple(slice(c_idx - p_idx, c_idx + (s_idx - p_idx))
                    for (c_idx, p_idx, s_idx) in zip(center, patch_half, patch_size)) for center in new_centers]
    return [np.squeeze(new_image[idx]) for idx in slices]


def get_mask_voxels(mask, size=None, rng=None):
    """base.py:310-331: C-order list of (x,y,z) tuples; optional shuffle + truncate.
    The reference shuffles with the unseeded stdlib RNG; `rng` (a random.Random) makes it reproducible."""
    indices = np.stack(np.nonzero(mask), axis=1)
    indices_list = [tuple(int(v) for v in idx) for idx in indices]
    if size is not None:
        (rng or random).shuffle(indices_list)
        indices_list = indices_list[:size]
    return indices_list


def zscore_inference(image):
    """base.py:358 - statistics over non-zero voxels, population std, applied in the image's own arithmetic."""
    nz = image[np.nonzero(image)]
    return (image - nz.mean()) / nz.std()


def zscore_training(image):
    """base.py:146 - (im.astype(float32) - mean) / std."""
    nz = image[np.nonzero(image)]
    return (image.astype(np.float32) - nz.mean()) / nz.std()


def atlas_vectors(atlas_image, centers):
    """base.py:388-394: per-centre 15-vector, all-zero rows become background (channel 14 = 1)."""
    v = np.stack([atlas_image[c[0], c[1], c[2]] for c in centers]).astype(np.float32)
    for index in range(v.shape[0]):
        if np.sum(v[index]) == 0:
            v[index, 14] = 1
    return v


def load_patch_batch(image, atlas_image, roi_mask, batch_size, patch_size=(32, 32), datatype=np.float32):
    """base.py:357-397 on arrays: yields (axial, coronal, saggital [B,1,p,p], atlas [B,15], centers)."""
    image_norm = zscore_inference(image)
    centers_all = get_mask_voxels(roi_mask.astype(bool))
    for i in range(0, len(centers_all), batch_size):
        centers = centers_all[i:i + batch_size]
        views = [np.stack([np.array(get_patches(image_norm, centers, patch_size, mode=m)).astype(datatype)], axis=1)
                 for m in ('axial', 'coronal', 'saggital')]
        yield views[0], views[1], views[2], atlas_vectors(atlas_image, centers), centers


def sample_training_voxels(labels, rng):
    """base.py:154,164: all voxels with 0<label<15, then an equal number of shuffled label==15 voxels."""
    pos = get_mask_voxels(np.logical_and(labels > 0, labels < 15))
    neg = get_mask_voxels(labels == 15, size=len(pos), rng=rng)
    return pos, neg


def subject_training_patches(image, labels, atlas_image, rng, patch_size=(32, 32)):
    """base.py:145-181 + 187-218 for one subject: ([x_ax, x_co, x_sa], y_axial, atlas) with [pos; neg] order.
    The training-path background fix-up of the reference is dead code (base.py:213-216 raises NameError only
    when a whole subject's matrix sums to 0); rows are returned unmodified here, as the reference does for
    any real subject."""
    im = zscore_training(image)
    pos, neg = sample_training_voxels(labels, rng)
    xs = []
    for m in ('axial', 'coronal', 'saggital'):
        xs.append(np.concatenate([np.array(get_patches(im, pos, patch_size, mode=m)),
                                  np.array(get_patches(im, neg, patch_size, mode=m))]))
    y_ax = np.concatenate([np.array(get_patches(labels, pos, patch_size, mode='axial')),
                           np.array(get_patches(labels, neg, patch_size, mode='axial'))])
    c = np.asarray(pos + neg)
    atl = atlas_image[c[:, 0], c[:, 1], c[:, 2]]
    return xs, y_ax, atl, pos + neg


def generate_training_set(x_axial, x_coronal, x_saggital, x_atlas, y, seed=None):
    """base.py:78-110. `seed` None = no permutation (randomize=False); else the single seed re-applied 5x."""
    xa = np.concatenate(x_axial, axis=0).astype('float32')
    xc = np.concatenate(x_coronal, axis=0).astype('float32')
    xs = np.concatenate(x_saggital, axis=0).astype('float32')
    xt = np.concatenate(x_atlas, axis=0).astype('float32')
    yt = np.concatenate(y, axis=0).astype('uint8')
    yt = yt[:, yt.shape[1] // 2, yt.shape[2] // 2]
    yt = np.squeeze(yt)
    yt[yt == 15] = 0
    if seed is not None:
        out = []
        for a in (xa, xc, xs, yt, xt):
            np.random.seed(seed)
            out.append(np.random.permutation(a))
        xa, xc, xs, yt, xt = out
    return (np.expand_dims(xa, 1), np.expand_dims(xc, 1), np.expand_dims(xs, 1), xt, yt)


def scatter_labels(shape, dtype, centers, y_pred):
    """base.py:415,430-431."""
    image = np.zeros(shape, dtype=dtype)
    x, y, z = np.stack(centers, axis=1)
    image[x, y, z] = y_pred
    return image


def post_process_segmentation(input_mask, atlas_mask):
    """base.py:460-480, including its quirk: labeled_comprehension is evaluated over np.unique(labels)
    (which contains the background component 0) and the argmax INDEX is then used as a component label."""
    filtered_mask = np.zeros_like(input_mask)
    for l in range(1, 15):
        th_label = input_mask == l
        labels, num_labels = ndimage.label(th_label)
        label_list = np.unique(labels)
        num_elements_by_lesion = ndimage.labeled_comprehension(np.logical_and(th_label, atlas_mask), labels,
                                                               label_list, np.sum, float, 0)
        argmax = np.argmax(num_elements_by_lesion)
        current_voxels = np.stack(np.where(labels == argmax), axis=1)
        filtered_mask[current_voxels[:, 0], current_voxels[:, 1], current_voxels[:, 2]] = l
    return filtered_mask
