"""
TEST INFRASTRUCTURE - NOT PRODUCT CODE (see oracle/np_oracle.py for the rules and the
"PARITY UNPINNED" statement, which applies here too).

Second, independent restatement of the reference net (/root/reference/cnn_cort/nets.py:170-246) on
torch-CPU ops: correlation kernels (F.conv2d) fed with spatially flipped weights, F.max_pool2d, autograd for
the training step.  Two uses:
  * float64: cross-checks oracle/np_oracle.py (different conv algorithm, different backward: autograd vs
    hand-written) - tests/test_oracle.py requires agreement to 1e-10.
  * float32, all host threads: the `cpu_baseline` leg of bench.py (kind "port": the reference's own Theano
    CPU path cannot run here - BASELINE.md section 3).
"""
import numpy as np
import torch
import torch.nn.functional as F

TOWERS = ('axial', 'coronal', 'saggital')
BN_EPS = 1e-4


def to_torch(params, dtype=torch.float64, requires_grad=False):
    out = {}
    for k, arrs in params.items():
        out[k] = [torch.tensor(np.asarray(a), dtype=dtype).requires_grad_(requires_grad) for a in arrs]
    return out


def _prelu(x, alpha):
    a = alpha.view((1, -1) + (1,) * (x.dim() - 2))
    return 0.5 * (1 + a) * x + 0.5 * (1 - a) * x.abs()


def _relu(x):
    return 0.5 * (x + x.abs())


def _conv_true(x, W):
    return F.conv2d(x, torch.flip(W, dims=(2, 3)))


def _maxpool2_theano(x):
    """2x2/2 max pool whose backward sends the whole output gradient to EVERY element equal to the window
    maximum (Theano MaxPoolGrad), unlike F.max_pool2d (first maximum only).  Matters for constant regions
    (zero-padded border patches), where windows tie at a positive value."""
    y = F.max_pool2d(x, 2)
    up = y.detach().repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    mask = (x.detach() == up).to(x.dtype)
    n, c, h, w = x.shape
    routed = ((x - x.detach()) * mask).reshape(n, c, h // 2, 2, w // 2, 2).sum(dim=(3, 5))
    return y.detach() + routed


def forward(P, ax, co, sa, atlas, train=False, masks=None, bn_out=None):
    """P: dict of torch tensors (to_torch). Inputs torch tensors. Returns probs [N,15]."""
    feats = []
    for t, x in zip(TOWERS, (ax, co, sa)):
        h = x
        for i in range(1, 6):
            W, = P['%s_ch_conv%d' % (t, i)]
            beta, gamma, mean, inv_std = P['%s_ch_conv%d_bn' % (t, i)]
            z = _conv_true(h, W)
            if train:
                mean = z.mean(dim=(0, 2, 3))
                inv_std = 1.0 / torch.sqrt(z.var(dim=(0, 2, 3), unbiased=False) + BN_EPS)
                if bn_out is not None:
                    bn_out['%s_ch_conv%d_bn' % (t, i)] = (mean.detach(), inv_std.detach())
            sh = (1, -1, 1, 1)
            h = (z - mean.view(sh)) * (gamma * inv_std).view(sh) + beta.view(sh)
            h = _prelu(_relu(h), P['%s_ch_prelu%d' % (t, i)][0])
            if i in (2, 4):
                h = _maxpool2_theano(h) if train else F.max_pool2d(h, 2)
        if train:
            h = h / 0.5 * masks['%s_l1drop' % t]
        W, b = P['%s_d1' % t]
        feats.append(_prelu(_relu(h.reshape(h.shape[0], -1) @ W + b), P['%s_prelu_d1' % t][0]))
    h = torch.cat(feats, dim=1)
    if train:
        h = h / 0.5 * masks['f1_drop']
    W, b = P['FC1']
    h = _prelu(_relu(h @ W + b), P['prelu_f1'][0])
    if train:
        h = h / 0.5 * masks['f2_drop']
    h = torch.cat([h, atlas], dim=1)
    W, b = P['fc_2']
    h = _prelu(_relu(h @ W + b), P['prelu_f2'][0])
    W, b = P['out_layer']
    return torch.softmax(h @ W + b, dim=1)


def train_loss_and_grads(params, ax, co, sa, atlas, y, masks, dtype=torch.float64):
    """Autograd version of np_oracle.train_forward_backward. Returns loss, grads (numpy), bn_batch."""
    P = to_torch(params, dtype, requires_grad=True)
    tt = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    bn = {}
    probs = forward(P, tt(ax), tt(co), tt(sa), tt(atlas), train=True,
                    masks={k: tt(v) for k, v in masks.items()}, bn_out=bn)
    yi = torch.tensor(np.asarray(y).astype(np.int64))
    loss = -torch.log(probs[torch.arange(probs.shape[0]), yi]).mean()
    loss.backward()
    grads = {k: [(a.grad.numpy() if a.grad is not None else np.zeros(tuple(a.shape))) for a in v]
             for k, v in P.items()}
    return float(loss.detach()), grads, {k: (m.numpy(), s.numpy()) for k, (m, s) in bn.items()}, probs.detach().numpy()


class FoldedCPUNet:
    """Deterministic float32 forward with BN folded into the convs, for CPU timing (bench.py cpu_baseline)."""

    def __init__(self, params, dtype=torch.float32):
        self.dtype = dtype
        P = to_torch(params, torch.float64)
        self.convs = {}
        for t in TOWERS:
            for i in range(1, 6):
                W, = P['%s_ch_conv%d' % (t, i)]
                beta, gamma, mean, inv_std = P['%s_ch_conv%d_bn' % (t, i)]
                s = gamma * inv_std
                self.convs[(t, i)] = ((torch.flip(W, dims=(2, 3)) * s.view(-1, 1, 1, 1)).to(dtype).contiguous(),
                                      (beta - mean * s).to(dtype))
        self.P = {k: [a.to(dtype) for a in v] for k, v in P.items()}

    @torch.no_grad()
    def forward(self, ax, co, sa, atlas):
        feats = []
        for t, x in zip(TOWERS, (ax, co, sa)):
            h = x
            for i in range(1, 6):
                W, b = self.convs[(t, i)]
                h = F.relu(F.conv2d(h, W, b))
                if i in (2, 4):
                    h = F.max_pool2d(h, 2)
            W, b = self.P['%s_d1' % t]
            feats.append(F.relu(h.reshape(h.shape[0], -1) @ W + b))
        W, b = self.P['FC1']
        h = F.relu(torch.cat(feats, 1) @ W + b)
        W, b = self.P['fc_2']
        h = F.relu(torch.cat([h, atlas], 1) @ W + b)
        W, b = self.P['out_layer']
        return torch.softmax(h @ W + b, 1)
