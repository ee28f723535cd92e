"""
Synthetic inputs for tests and benchmarks (no T1 image or atlas ships with the reference:
/root/reference/.MISSING_LARGE_BLOBS:1-6).  Shapes follow SURVEY.md section 8(d).
"""
import numpy as np
from scipy import ndimage


def patch_batch(n, seed=0):
    """BASELINE config 2 inputs: ax/co/sa ~ N(0,1) [n,1,32,32] f32 (seeds s, s+1, s+2),
    atlas ~ Dirichlet(1_15) [n,15] f32 (seed s+3)."""
    views = [np.random.RandomState(seed + i).standard_normal((n, 1, 32, 32)).astype(np.float32) for i in range(3)]
    atlas = np.random.RandomState(seed + 3).dirichlet(np.ones(15), n).astype(np.float32)
    return views[0], views[1], views[2], atlas


def smooth_volume(shape=(96, 96, 96), sigma=4.0, seed=0, dtype=np.float32, as_int=False):
    """Gaussian-filtered noise, rescaled to an MR-like positive intensity range with a zero background shell."""
    rng = np.random.RandomState(seed)
    v = ndimage.gaussian_filter(rng.standard_normal(shape), sigma)
    v = (v - v.min()) / (v.max() - v.min())
    v = 50.0 + 900.0 * v
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing='ij'))
    v[(g ** 2).sum(0) > 1.0] = 0.0           # outside the "head": exactly zero (ignored by the z-score)
    if as_int:
        return np.round(v).astype(np.int16)
    return v.astype(dtype)


def ellipsoid_mask(shape, radii=(0.35, 0.3, 0.3), center=(0.0, 0.0, 0.0)):
    g = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing='ij')
    return sum(((gi - c) / r) ** 2 for gi, c, r in zip(g, center, radii)) <= 1.0


def toy_atlas(shape, seed=1, n_struct=14):
    """15-channel prior [X,Y,Z,15]: 14 soft blobs inside the ROI + background channel 14; rows sum to 1
    inside the blobs' support and are ALL ZERO far away (exercises the zero-row -> background fix-up,
    /root/reference/cnn_cort/base.py:391-394)."""
    rng = np.random.RandomState(seed)
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing='ij'), axis=-1)
    a = np.zeros(tuple(shape) + (15,), np.float32)
    for c in range(n_struct):
        ctr = rng.uniform(-0.25, 0.25, 3)
        rad = rng.uniform(0.08, 0.18)
        d2 = ((g - ctr) ** 2).sum(-1)
        a[..., c] = np.exp(-d2 / (2 * rad ** 2)) * (d2 < (2.5 * rad) ** 2)
    s = a[..., :14].sum(-1)
    near = s > 1e-3
    a[..., 14] = np.where(near, np.clip(1.0 - s, 0, None), 0.0)
    tot = a.sum(-1, keepdims=True)
    a = np.where(tot > 0, a / np.maximum(tot, 1e-12), 0.0).astype(np.float32)
    a[~near] = 0.0
    return a


def toy_labels(shape, atlas, seed=2):
    """GT volume in the reference's convention: 1..14 structures, 15 boundary negatives, 0 rest."""
    lab = np.zeros(shape, np.uint8)
    fg = atlas[..., :14].max(-1) > 0.5
    lab[fg] = (atlas[..., :14].argmax(-1)[fg] + 1).astype(np.uint8)
    ring = ndimage.binary_dilation(fg, iterations=3) & ~fg
    lab[ring] = 15
    return lab


def steer_matrix():
    """[15][15]: row k = the prior vector that makes the shipped net answer class (k + 1) for k < 14 and class 0 (background) for
    k = 14, i.e. prior channel k steers towards the structure the reference's atlas stores in channel k (channel 14 = background,
    SURVEY F10).  Vectors: cnn_cort/steer15.json, written by tools/make_steer.py."""
    import json
    import os
    v = np.asarray(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'steer15.json')))['vectors'], np.float32)
    return np.concatenate([v[1:], v[:1]], axis=0)


def steered_atlas(shape, seed=1, radius=(0.10, 0.20)):
    """Prior [X,Y,Z,15] f32 of a synthetic subject on which the shipped net shows EVERY class at volume level - what smooth volumes
    with probability-like priors never do (classes 0 and 2 only).  Fourteen blobs as in toy_atlas; inside
    blob k the prior row is the vector that makes class k + 1 win (steer_matrix), fading linearly into the background vector
    towards the blob's rim (mixtures: unsaturated probabilities and near ties there), the background vector between the blobs, and
    all-zero rows far away (the zero-row -> background rule of base.py:391-394).  Returns (prior, structure map: 0 = none,
    1..14 = blob) - the map's non-zero voxels are what a registered sub-cortical mask would cover."""
    rng = np.random.RandomState(seed)
    M = steer_matrix()
    axes = [np.linspace(-1, 1, s) for s in shape]
    best = np.zeros(shape, np.float32)
    who = np.zeros(shape, np.int8)
    for c in range(14):
        ctr = rng.uniform(-0.25, 0.25, 3)
        rad = rng.uniform(*radius)
        ex, ey, ez = [np.exp(-(a - m) ** 2 / (2 * rad ** 2)).astype(np.float32) for a, m in zip(axes, ctr)]   # a Gaussian blob is separable
        f = ex[:, None, None] * ey[None, :, None] * ez[None, None, :]
        take = f > best
        best[take] = f[take]
        who[take] = c + 1
    inside = best > 0.2
    prior = np.zeros(tuple(shape) + (15,), np.float32)
    near = np.flatnonzero(best >= 0.02)                        # rows elsewhere stay all zero
    b = best.ravel()[near]
    wgt = np.clip((b - 0.2) / 0.3, 0.0, 1.0).astype(np.float32)[:, None]
    prior.reshape(-1, 15)[near] = wgt * M[who.ravel()[near] - 1] + (1.0 - wgt) * M[14]
    return prior, np.where(inside, who, 0).astype(np.uint8)
