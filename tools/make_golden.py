#!/usr/bin/env python3
"""
Generates the committed golden fixtures under tests/golden/ from the float64 oracle (oracle/np_oracle.py)
and the shipped miccai2012_v1 weights.  The reference itself cannot run here (Python 2 + Theano/Lasagne,
SURVEY.md section 8c), so these vectors pin OUR oracle against regressions and give the GPU tests fixed,
discriminating inputs; they are not outputs of the reference ("parity unpinned").

  g1_random.npz   64 N(0,1) tri-view patches + Dirichlet atlas -> logits, probs; per-layer trace of 2 patches
  g2_volume.npz   256 patches cut by the literal get_patches restatement from a 48^3 synthetic volume
                  (border centres included -> zero padding) -> logits, probs
  g3_classes.npz  120 inputs, 8 per class, found by gradient ascent so every class 0..14 is an argmax
  g4_ties.npz     32 inputs whose top-2 probabilities are within 1e-3 of each other
  g5_train.npz    one float64 training step on 32 samples: loss, per-array gradient digests, Adam digests
  g6_toy.npz      24^3 toy volume: centres, atlas vectors, patch digests, training-set assembly
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'sub-cortical_segmentation_amd'))
from cnn_cort import params as P, synth  # noqa: E402
from oracle import np_oracle as O, torch_ref as T, patches_ref as R  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
W = P.load_params(os.path.join(ROOT, 'sub-cortical_segmentation_amd', 'nets', 'miccai2012_v1', 'miccai2012_v1'))


def logits_probs(ax, co, sa, at):
    tr = {}
    p = O.forward(W, ax, co, sa, at, trace=tr)
    return tr['logits'], p, tr


def save(name, **kw):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **kw)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def g1():
    ax, co, sa, at = synth.patch_batch(64, seed=100)
    z, p, _ = logits_probs(ax, co, sa, at)
    _, _, tr = logits_probs(ax[:2], co[:2], sa[:2], at[:2])
    trace = {'trace_' + k: v.astype(np.float32) for k, v in tr.items()}
    save('g1_random.npz', ax=ax, co=co, sa=sa, atlas=at, logits=z, probs=p, **trace)


def volume_patches(n=256, shape=(48, 48, 48), seed=7):
    vol = synth.smooth_volume(shape, sigma=3.0, seed=seed)
    atlas = synth.toy_atlas(shape, seed=seed + 1)
    rng = np.random.RandomState(seed + 2)
    centers = [tuple(int(v) for v in rng.randint(0, s, 3)) for s in [shape] * (n - 16)]
    # corners / faces: every view needs zero padding somewhere
    edge = [0, shape[0] - 1]
    centers += [(x, y, z) for x in edge for y in edge for z in edge]
    centers += [(0, 24, 24), (24, 0, 24), (24, 24, 0), (47, 24, 24), (24, 47, 24), (24, 24, 47), (1, 1, 1), (46, 46, 46)]
    centers = centers[:n]
    vn = R.zscore_inference(vol)
    views = [np.stack([np.array(R.get_patches(vn, centers, (32, 32), mode=m)).astype(np.float32)], axis=1)
             for m in ('axial', 'coronal', 'saggital')]
    at = R.atlas_vectors(atlas, centers)
    return vol, atlas, np.asarray(centers, np.int32), views, at


def g2():
    vol, atlas, centers, (ax, co, sa), at = volume_patches()
    z, p, _ = logits_probs(ax, co, sa, at)
    save('g2_volume.npz', volume=vol, atlas4d=atlas, centers=centers, ax=ax, co=co, sa=sa, atlas=at, logits=z, probs=p)


def ascend(target_fn, x0, steps=300, lr=0.05, free_atlas=False):
    """Gradient ascent on the three views (and optionally on the atlas row, kept on the simplex through a
    softmax parametrisation) with the float64 torch restatement."""
    Pt = T.to_torch(W)
    xs = [torch.tensor(np.asarray(a, np.float64), requires_grad=True) for a in x0[:3]]
    al = torch.tensor(np.log(np.asarray(x0[3], np.float64) + 1e-3), requires_grad=free_atlas)
    opt = torch.optim.Adam(xs + ([al] if free_atlas else []), lr=lr)
    for _ in range(steps):
        opt.zero_grad()
        at = torch.softmax(al, dim=1) if free_atlas else torch.tensor(np.asarray(x0[3], np.float64))
        probs = T.forward(Pt, xs[0], xs[1], xs[2], at)
        loss = target_fn(torch.log(probs + 1e-300))
        loss.backward()
        opt.step()
    at = torch.softmax(al, dim=1).detach().numpy().astype(np.float32) if free_atlas else np.asarray(x0[3], np.float32)
    return [x.detach().numpy().astype(np.float32) for x in xs] + [at]


def g3():
    cls = np.repeat(np.arange(15), 8)
    ax, co, sa, at = synth.patch_batch(120, seed=300)
    # atlas channel k is the prior of label k+1, channel 14 the background prior (label 0) - SURVEY F10
    at = (0.1 * at + 0.9 * np.eye(15, dtype=np.float32)[(cls - 1) % 15]).astype(np.float32)
    ct = torch.tensor(cls)
    ax, co, sa, at = ascend(lambda lp: -lp[torch.arange(120), ct].sum(), (ax, co, sa, at), steps=500, lr=0.1,
                            free_atlas=True)
    z, p, _ = logits_probs(ax, co, sa, at)
    got = p.argmax(1)
    print('g3 classes hit:', np.bincount(got, minlength=15))
    assert (np.bincount(got, minlength=15) >= 2).all(), 'class-spanning set incomplete'
    save('g3_classes.npz', ax=ax, co=co, sa=sa, atlas=at, logits=z, probs=p, target=cls)


def g4():
    n = 32
    ax, co, sa, at = synth.patch_batch(n, seed=400)
    a = np.random.RandomState(5).randint(0, 15, n)
    b = (a + 1 + np.random.RandomState(6).randint(0, 14, n)) % 15
    at = (0.2 * at + 0.4 * np.eye(15, dtype=np.float32)[(a - 1) % 15] + 0.4 * np.eye(15, dtype=np.float32)[(b - 1) % 15]).astype(np.float32)
    ta, tb = torch.tensor(a), torch.tensor(b)
    idx = torch.arange(n)

    def tie(lp):            # cross-entropy to a 50/50 target on (a, b): minimised at p_a = p_b = 0.5
        return -(0.5 * lp[idx, ta] + 0.5 * lp[idx, tb]).sum()
    ax, co, sa, at = ascend(tie, (ax, co, sa, at), steps=700, lr=0.05, free_atlas=True)
    z, p, _ = logits_probs(ax, co, sa, at)
    srt = np.sort(p, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    print('g4 top-2 margins: min %.2e median %.2e max %.2e' % (margin.min(), np.median(margin), margin.max()))
    save('g4_ties.npz', ax=ax, co=co, sa=sa, atlas=at, logits=z, probs=p, margin=margin)


def digest(a):
    a = np.asarray(a, np.float64).ravel()
    return np.array([a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()] + list(a[:5]) + [0.0] * max(0, 5 - a.size))[:8]


def g5():
    n = 32
    _, _, _, (ax, co, sa), at = volume_patches(n=n, seed=44)
    rng = np.random.RandomState(45)
    y = rng.randint(0, 15, n).astype(np.uint8)
    masks = {k: (rng.rand(n, 60, 3, 3) < 0.5).astype(np.uint8) for k in ('axial_l1drop', 'coronal_l1drop', 'saggital_l1drop')}
    masks['f1_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    masks['f2_drop'] = (rng.rand(n, 540) < 0.5).astype(np.uint8)
    loss, grads, bn, probs = O.train_forward_backward(W, ax, co, sa, at, y, masks)
    state = {}
    newp = O.adam_step(W, grads, state, trainable=P.is_trainable)
    newp = O.bn_running_update(newp, bn)
    names = [k for k, s in P.LAYER_SPEC if s]
    gd = np.stack([digest(g) for k in names for g in grads[k]])
    pd = np.stack([digest(np.asarray(a) - np.asarray(b, np.float64)) for k in names for a, b in zip(newp[k], W[k])])
    save('g5_train.npz', ax=ax, co=co, sa=sa, atlas=at, y=y, loss=np.array(loss), probs=probs,
         grad_digest=gd, delta_digest=pd, **{'mask_' + k: v for k, v in masks.items()})


def g6():
    shape = (24, 24, 24)
    vol = synth.smooth_volume(shape, sigma=2.0, seed=50, as_int=True)       # integer T1 -> float64 z-score path
    atlas = synth.toy_atlas(shape, seed=51)
    labels = synth.toy_labels(shape, atlas, seed=52)
    roi = np.zeros(shape, bool)
    roi[6:18, 5:19, 0:9] = True                                             # touches the z=0 face
    out = {'volume': vol, 'atlas4d': atlas, 'labels': labels, 'roi': roi.astype(np.uint8)}
    batches = list(R.load_patch_batch(vol, atlas, roi, batch_size=500))
    out['n_batches'] = np.array(len(batches))
    out['centers'] = np.concatenate([np.asarray(b[4], np.int32) for b in batches])
    out['atlas_vec'] = np.concatenate([b[3] for b in batches])
    for vi, nm in enumerate(('ax', 'co', 'sa')):
        allp = np.concatenate([b[vi] for b in batches])
        out['sum_' + nm] = allp.astype(np.float64).sum(axis=(1, 2, 3))
        out['first_' + nm] = allp[:4]
        out['last_' + nm] = allp[-4:]
    # training-set assembly (seeded sampler + seeded permutation)
    xs, y_ax, atl, cent = R.subject_training_patches(vol, labels, atlas, random.Random(1234))
    xa, xc, xsg, xt, yt = R.generate_training_set([xs[0]], [xs[1]], [xs[2]], [atl], [y_ax], seed=777)
    out.update(train_centers=np.asarray(cent, np.int32), train_y=yt, train_atlas=xt,
               train_sum_ax=xa.astype(np.float64).sum(axis=(1, 2, 3)),
               train_sum_co=xc.astype(np.float64).sum(axis=(1, 2, 3)),
               train_sum_sa=xsg.astype(np.float64).sum(axis=(1, 2, 3)),
               train_first_ax=xa[:2])
    # post-processing fixture: a noisy label map
    rng = np.random.RandomState(53)
    seg = np.where(rng.rand(*shape) < 0.08, rng.randint(1, 15, shape), 0).astype(np.int16)
    seg[labels > 0] = np.where(labels[labels > 0] < 15, labels[labels > 0], 0)
    amask = (atlas[..., :13].sum(-1) > 0)
    out.update(pp_in=seg, pp_mask=amask.astype(np.uint8), pp_out=R.post_process_segmentation(seg, amask))
    save('g6_toy.npz', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6']
    for w in which:
        globals()[w]()
