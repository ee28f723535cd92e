#!/usr/bin/env python3
"""
Writes sub-cortical_segmentation_amd/cnn_cort/steer15.json: fifteen "prior" vectors a_c (15 floats each) such that, with the
shipped miccai2012_v1 weights and patches cut from cnn_cort.synth.smooth_volume volumes, the net's argmax is class c when the
atlas input (in4) of a voxel is a_c.

Why: smooth synthetic volumes with probability-like priors only ever produce classes 0 and 2, so no
volume-level test or bench line would see test_scan -> post-process -> per-class Dice with 14 live structures.  The prior enters the
net linearly (fc_2's rows 540..554, /root/reference/cnn_cort/nets.py:222-228), and the prior FILE is caller data of any float
value, so a synthetic subject whose prior holds a_c inside blob c is a legitimate input that makes every class win somewhere
(cnn_cort.synth.steered_atlas).  The vectors come from gradient ascent on the torch-CPU restatement (oracle/torch_ref.py) over
patches of three synthetic volumes and are checked on a fourth; they are data, committed with this script.

  python tools/make_steer.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'sub-cortical_segmentation_amd')
sys.path[:0] = [ROOT, PKG]

from cnn_cort import params as P, synth           # noqa: E402
from oracle import patches_ref as R, torch_ref as T   # noqa: E402


def fc1_features(net, shape, seed, n, rng):
    vol = R.zscore_inference(synth.smooth_volume(shape, seed=seed, as_int=True)).astype(np.float32)
    cs = [tuple(int(rng.randint(s // 4, s - s // 4)) for s in shape) for _ in range(n)]
    views = [torch.from_numpy(np.stack([np.array(R.get_patches(vol, cs, (32, 32), mode=m)).astype(np.float32)], axis=1))
             for m in ('axial', 'coronal', 'saggital')]
    with torch.no_grad():
        feats = []
        for t, x in zip(T.TOWERS, views):
            h = x
            for i in range(1, 6):
                W, b = net.convs[(t, i)]
                h = F.relu(F.conv2d(h, W, b))
                if i in (2, 4):
                    h = F.max_pool2d(h, 2)
            W, b = net.P['%s_d1' % t]
            feats.append(F.relu(h.reshape(h.shape[0], -1) @ W + b))
        W, b = net.P['FC1']
        return F.relu(torch.cat(feats, 1) @ W + b)


def main():
    torch.manual_seed(0)
    w = P.load_params(os.path.join(PKG, 'nets', 'miccai2012_v1', 'miccai2012_v1'))
    net = T.FoldedCPUNet(w)
    rng = np.random.RandomState(0)
    train = torch.cat([fc1_features(net, sh, sd, 96, rng) for sh, sd in (((64, 64, 64), 3), ((72, 64, 56), 11), ((96, 96, 96), 0))])
    held = fc1_features(net, (80, 80, 64), 5, 128, rng)
    W2, b2 = net.P['fc_2']
    Wo, bo = net.P['out_layer']

    def logits(r1, a):
        return F.relu(torch.cat([r1, a.expand(r1.shape[0], 15)], 1) @ W2 + b2) @ Wo + bo

    vecs, hits = [], []
    for c in range(15):
        a = torch.zeros(15, requires_grad=True)
        opt = torch.optim.Adam([a], lr=0.1)
        for _ in range(500):
            loss = F.cross_entropy(logits(train, a), torch.full((train.shape[0],), c)) + 1e-3 * (a ** 2).sum()
            opt.zero_grad()
            loss.backward()
            opt.step()
        with torch.no_grad():
            hit = float((logits(held, a).argmax(1) == c).float().mean())
        vecs.append([float(np.float32(v)) for v in a.detach().numpy()])
        hits.append(hit)
        print('class %2d: held-out hit rate %.2f, max |a| %.2f' % (c, hit, max(abs(v) for v in vecs[-1])))
    assert min(hits) > 0.5, 'a class cannot be steered'
    out = os.path.join(PKG, 'cnn_cort', 'steer15.json')
    json.dump({'note': 'tools/make_steer.py: prior vectors that make class c the argmax on smooth synthetic volumes (shipped weights)',
               'held_out_hit_rate': hits, 'vectors': vecs}, open(out, 'w'), indent=0)
    print('wrote', out)


if __name__ == '__main__':
    main()
