"""
TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (sub-cortical_segmentation_amd/) never does.

float64 NumPy restatement of the reference's voxelwise CNN (the "oracle").

PARITY UNPINNED: the arithmetic of the reference lives in un-vendored third-party packages
(Theano==0.9.0, Lasagne==0.2.dev1, nolearn==0.6.0 - /root/reference/requirements.txt:32,9,16) which are
absent from /root/reference and not installable here, the reference's own .py files are Python-2-only,
and the reference ships no tests, golden vectors or fixtures.  This restatement therefore follows
 (i) the topology and hyper-parameters written at /root/reference/cnn_cort/nets.py:159-246,
 (ii) the published semantics of the Lasagne/Theano ops that code calls (restated below from the
      upstream documentation), and
 (iii) the tensor shapes / layer order of the shipped weights nets/miccai2012_v1/miccai2012_v1.pkl,
and is cross-checked against an independent torch-CPU restatement (oracle/torch_ref.py), not against the
reference itself.

Semantics restated:
  * batch_norm(Conv2DLayer(...)) (nets.py:171): the conv loses its bias, a BatchNormLayer follows, and the
    conv's default `rectify` is applied after the BN; prelu(...) wraps that.  Order: conv -> BN -> ReLU -> PReLU.
  * Conv2DLayer default flip_filters=True, pad='valid', stride 1: a TRUE convolution,
    out[n,o,y,x] = sum_{c,i,j} in[n,c,y+i,x+j] * W[o,c,2-i,2-j].
  * BatchNormLayer params [beta, gamma, mean, inv_std]; deterministic: (x-mean)*(gamma*inv_std)+beta;
    training: batch mean / biased variance over (N,H,W), inv_std = 1/sqrt(var+1e-4), running averages
    updated with alpha = 0.1.
  * rectify = theano.tensor.nnet.relu(x) = 0.5*(x+|x|).
  * ParametricRectifierLayer: 0.5*(1+alpha)*x + 0.5*(1-alpha)*|x|, alpha shared over all axes but 1.
  * MaxPool2DLayer(pool_size=2): 2x2 window, stride 2, ignore_border (inputs here are even).
  * DropoutLayer(p): identity when deterministic; else x/(1-p)*mask.
  * DenseLayer: flatten trailing axes (c,h,w order), x.W + b, W stored (in,out); default rectify.
  * out_layer nonlinearity softmax (row-wise, max-subtracted); loss mean(-log p[y]) (nets.py:235);
    update lasagne.updates.adam(lr=1e-3, b1=.9, b2=.999, eps=1e-8) (nets.py:236-237).
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

TOWERS = ('axial', 'coronal', 'saggital')
BN_EPS = 1e-4          # lasagne.layers.BatchNormLayer default epsilon
BN_ALPHA = 0.1         # ... default alpha (running-average rate)


# ----------------------------------------------------------------------------------- primitives
def conv3x3_true(x, W):
    """Valid true convolution (flip_filters=True). x [N,C,H,W], W [O,C,3,3] -> [N,O,H-2,W-2]."""
    Wf = W[:, :, ::-1, ::-1]
    win = sliding_window_view(x, (3, 3), axis=(2, 3))            # [N,C,Ho,Wo,3,3]
    out = np.tensordot(win, Wf, axes=([1, 4, 5], [1, 2, 3]))     # [N,Ho,Wo,O]
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2))


def conv3x3_true_loops(x, W):
    """Same op as conv3x3_true, written as explicit tap loops (independent cross-check)."""
    N, C, H, Wd = x.shape
    O = W.shape[0]
    out = np.zeros((N, O, H - 2, Wd - 2), dtype=x.dtype)
    for i in range(3):
        for j in range(3):
            # true convolution: tap (i,j) of the input window meets W[..., 2-i, 2-j]
            out += np.einsum('nchw,oc->nohw', x[:, :, i:i + H - 2, j:j + Wd - 2], W[:, :, 2 - i, 2 - j])
    return out


def bn_apply(x, beta, gamma, mean, inv_std):
    sh = (1, -1, 1, 1)
    return (x - mean.reshape(sh)) * (gamma * inv_std).reshape(sh) + beta.reshape(sh)


def relu(x):
    return 0.5 * (x + np.abs(x))


def prelu(x, alpha):
    a = alpha.reshape((1, -1) + (1,) * (x.ndim - 2))
    return 0.5 * (1 + a) * x + 0.5 * (1 - a) * np.abs(x)


def maxpool2(x):
    N, C, H, W = x.shape
    return x.reshape(N, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))


def softmax(z):
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def _p64(params):
    return {k: [np.asarray(a, np.float64) for a in v] for k, v in params.items()}


# ----------------------------------------------------------------------------------- inference
def tower_forward(params, tower, x, trace=None):
    """One tower (nets.py:170-180), deterministic. x [N,1,32,32] -> [N,180]."""
    h = x
    for i in range(1, 6):
        W, = params['%s_ch_conv%d' % (tower, i)]
        beta, gamma, mean, inv_std = params['%s_ch_conv%d_bn' % (tower, i)]
        alpha, = params['%s_ch_prelu%d' % (tower, i)]
        h = prelu(relu(bn_apply(conv3x3_true(h, W), beta, gamma, mean, inv_std)), alpha)
        if trace is not None:
            trace['%s_conv%d' % (tower, i)] = h
        if i in (2, 4):
            h = maxpool2(h)
            if trace is not None:
                trace['%s_pool%d' % (tower, i // 2)] = h
    W, b = params['%s_d1' % tower]
    alpha, = params['%s_prelu_d1' % tower]
    h = prelu(relu(h.reshape(h.shape[0], -1) @ W + b), alpha)
    if trace is not None:
        trace['%s_d1' % tower] = h
    return h


def forward(params, ax, co, sa, atlas, trace=None, chunk=64):
    """Deterministic forward of the whole net (nets.py:170-231) in float64.
    ax/co/sa [N,1,32,32], atlas [N,15] -> probs [N,15] float64."""
    P = _p64(params)
    N = ax.shape[0]
    outs = []
    for s in range(0, N, chunk):
        tr = {} if trace is not None else None
        sl = slice(s, min(N, s + chunk))
        feats = [tower_forward(P, t, np.asarray(x[sl], np.float64), tr) for t, x in zip(TOWERS, (ax, co, sa))]
        h = np.concatenate(feats, axis=1)                                        # elem_channels
        W, b = P['FC1']
        h = prelu(relu(h @ W + b), P['prelu_f1'][0])
        if tr is not None:
            tr['FC1'] = h
        h = np.concatenate([h, np.asarray(atlas[sl], np.float64)], axis=1)       # elem_channels2
        W, b = P['fc_2']
        h = prelu(relu(h @ W + b), P['prelu_f2'][0])
        if tr is not None:
            tr['fc_2'] = h
        W, b = P['out_layer']
        z = h @ W + b
        if tr is not None:
            tr['logits'] = z
            for k, v in tr.items():
                trace.setdefault(k, []).append(v)
        outs.append(softmax(z))
    if trace is not None:
        for k in list(trace.keys()):
            trace[k] = np.concatenate(trace[k], axis=0)
    return np.concatenate(outs, axis=0) if outs else np.zeros((0, 15))


def predict(params, ax, co, sa, atlas):
    """nolearn NeuralNet.predict: argmax of predict_proba (first maximum)."""
    return np.argmax(forward(params, ax, co, sa, atlas), axis=1)


# ----------------------------------------------------------------------------------- training
def _conv_bwd(x, W, dz):
    """Backward of conv3x3_true: returns (dx, dW)."""
    N, C, H, Wd = x.shape
    dx = np.zeros_like(x)
    dW = np.zeros_like(W)
    Ho, Wo = H - 2, Wd - 2
    for i in range(3):
        for j in range(3):
            xs = x[:, :, i:i + Ho, j:j + Wo]
            dW[:, :, 2 - i, 2 - j] = np.einsum('nohw,nchw->oc', dz, xs)
            dx[:, :, i:i + Ho, j:j + Wo] += np.einsum('nohw,oc->nchw', dz, W[:, :, 2 - i, 2 - j])
    return dx, dW


def _relu_bwd(x, g):
    return g * 0.5 * (1 + np.sign(x))


def _prelu_bwd(x, alpha, g):
    a = alpha.reshape((1, -1) + (1,) * (x.ndim - 2))
    dx = g * (0.5 * (1 + a) + 0.5 * (1 - a) * np.sign(x))
    axes = (0,) + tuple(range(2, x.ndim))
    dalpha = (g * (0.5 * x - 0.5 * np.abs(x))).sum(axis=axes)
    return dx, dalpha


def _pool_bwd(x, y, g):
    """Theano MaxPoolGrad: the gradient goes to every element equal to the window maximum."""
    up = lambda t: t.repeat(2, axis=2).repeat(2, axis=3)
    return (x == up(y)) * up(g)


def train_forward_backward(params, ax, co, sa, atlas, y, masks, debug=None, gates=None):
    """One training forward/backward in float64 (deterministic=False semantics).

    masks: dict with 0/1 arrays 'axial_l1drop','coronal_l1drop','saggital_l1drop' [N,60,3,3],
           'f1_drop' [N,540], 'f2_drop' [N,540]   (Bernoulli(0.5) keep masks; x*mask/(1-p)).
    gates: optional forced decisions of the BACKWARD pass, for comparing with a float32 implementation whose pre-activations
           within rounding of zero (or of a tied pooling window) fall on the other side: {'relu': {name: bool array shaped like
           the layer's pre-activation}, 'pool': {name: bool array shaped like the pooling INPUT, True where the window's
           gradient goes}}; names '<tower>_conv<i>', '<tower>_d1', 'FC1', 'fc_2'.  The forward pass is not touched (a
           pre-activation of +-1e-7 contributes nothing either way).  With `debug`, debug['forced'] lists per name how many
           decisions differ from the oracle's own and the largest |pre-activation| (relative to the layer's largest) among them.
    Returns loss, grads {layer: [arrays]} (zeros for BN mean/inv_std), bn_batch {layer: (mean, inv_std)},
    probs [N,15].
    """
    forced = {}

    def act_bwd(name, pre, r, alpha, g):
        """Backward of prelu(relu(pre)): -> (d pre, d alpha).  Forced gate: PReLU on a ReLU output is the identity wherever the
        gate is open, whatever value the oracle's own forward pass holds there (its r may be 0 where the float32 step saw
        +1e-8: the PReLU derivative AT zero, 0.5 (1 + alpha), must not enter)."""
        fg = (gates or {}).get('relu', {}).get(name)
        gp, da = _prelu_bwd(r, alpha, g)
        if fg is None:
            return _relu_bwd(pre, gp), da
        fg = np.asarray(fg, bool).reshape(pre.shape)
        diff = fg != (pre > 0)
        forced['relu:' + name] = (int(diff.sum()), float(np.abs(pre[diff]).max() / max(np.abs(pre).max(), 1e-300)) if diff.any() else 0.0)
        return g * fg, da

    def pool_bwd(name, x, yv, g):
        fg = (gates or {}).get('pool', {}).get(name)
        up = lambda t: t.repeat(2, axis=2).repeat(2, axis=3)
        if fg is None:
            return _pool_bwd(x, yv, g)
        fg = np.asarray(fg, bool).reshape(x.shape)
        diff = (fg != (x == up(yv))) & (up(yv) > 0)          # (windows without a positive value: the ReLU gate closes them anyway)
        forced['pool:' + name] = (int(diff.sum()), float(np.abs(x - up(yv))[diff].max() / max(np.abs(x).max(), 1e-300)) if diff.any() else 0.0)
        return up(g) * fg

    P = _p64(params)
    N = ax.shape[0]
    grads = {k: [np.zeros_like(a) for a in v] for k, v in P.items()}
    bn_batch = {}
    cache = {}
    feats = []
    for t, x in zip(TOWERS, (ax, co, sa)):
        h = np.asarray(x, np.float64)
        c = []
        for i in range(1, 6):
            W, = P['%s_ch_conv%d' % (t, i)]
            beta, gamma, _, _ = P['%s_ch_conv%d_bn' % (t, i)]
            alpha, = P['%s_ch_prelu%d' % (t, i)]
            z = conv3x3_true(h, W)
            mean = z.mean(axis=(0, 2, 3))
            inv_std = 1.0 / np.sqrt(z.var(axis=(0, 2, 3)) + BN_EPS)
            bn_batch['%s_ch_conv%d_bn' % (t, i)] = (mean, inv_std)
            xhat = (z - mean.reshape(1, -1, 1, 1)) * inv_std.reshape(1, -1, 1, 1)
            yb = xhat * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
            r = relu(yb)
            p = prelu(r, alpha)
            rec = dict(x=h, W=W, xhat=xhat, inv_std=inv_std, gamma=gamma, yb=yb, r=r, alpha=alpha, p=p)
            h = p
            if i in (2, 4):
                h = maxpool2(p)
                rec['pooled'] = h
            c.append(rec)
        m = np.asarray(masks['%s_l1drop' % t], np.float64)
        hd = h / 0.5 * m
        W, b = P['%s_d1' % t]
        pre = hd.reshape(N, -1) @ W + b
        r = relu(pre)
        f = prelu(r, P['%s_prelu_d1' % t][0])
        cache[t] = dict(convs=c, hd=hd, pre=pre, r=r)
        feats.append(f)
    cat = np.concatenate(feats, axis=1)
    m1 = np.asarray(masks['f1_drop'], np.float64)
    cat_d = cat / 0.5 * m1
    W1, b1 = P['FC1']
    pre1 = cat_d @ W1 + b1
    r1 = relu(pre1)
    h1 = prelu(r1, P['prelu_f1'][0])
    m2 = np.asarray(masks['f2_drop'], np.float64)
    h1d = h1 / 0.5 * m2
    cat2 = np.concatenate([h1d, np.asarray(atlas, np.float64)], axis=1)
    W2, b2 = P['fc_2']
    pre2 = cat2 @ W2 + b2
    r2 = relu(pre2)
    h2 = prelu(r2, P['prelu_f2'][0])
    W3, b3 = P['out_layer']
    logits = h2 @ W3 + b3
    probs = softmax(logits)
    yi = np.asarray(y).astype(np.int64)
    # mean categorical cross-entropy (nets.py:235) from the LOGITS: logsumexp(z) - z[y].  -log(softmax(z)[y]) is the same number
    # until a probability underflows (a logit gap beyond ~745 in float64), where it turns into inf and every comparison
    # against it becomes vacuous; `probs` stays what the backward pass uses.
    zmax = logits.max(axis=1)
    lse = zmax + np.log(np.exp(logits - zmax[:, None]).sum(axis=1))
    loss = float((lse - logits[np.arange(N), yi]).mean())

    # ---- backward
    dz = probs.copy()
    dz[np.arange(N), yi] -= 1.0
    dz /= N
    grads['out_layer'] = [h2.T @ dz, dz.sum(0)]
    g = dz @ W3.T
    g, da = act_bwd('fc_2', pre2, r2, P['prelu_f2'][0], g)
    grads['prelu_f2'] = [da]
    grads['fc_2'] = [cat2.T @ g, g.sum(0)]
    if debug is not None:
        debug.update(dpre2=g.copy(), cat2=cat2, r1=r1, cat_d=cat_d,
                     relu_margin=min(float(np.abs(pre1).min()), float(np.abs(pre2).min()),
                                     min(float(np.abs(cache[t]['pre']).min()) for t in TOWERS)))
    g = (g @ W2.T)[:, :540]                                  # the atlas input takes no gradient
    g = g / 0.5 * m2
    g, da = act_bwd('FC1', pre1, r1, P['prelu_f1'][0], g)
    grads['prelu_f1'] = [da]
    grads['FC1'] = [cat_d.T @ g, g.sum(0)]
    if debug is not None:
        debug.update(dpre1=g.copy())
    g = (g @ W1.T) / 0.5 * m1
    for ti, t in enumerate(TOWERS):
        ch = cache[t]
        gt = g[:, 180 * ti:180 * (ti + 1)]
        gt, da = act_bwd('%s_d1' % t, ch['pre'], ch['r'], P['%s_prelu_d1' % t][0], gt)
        grads['%s_prelu_d1' % t] = [da]
        W, b = P['%s_d1' % t]
        grads['%s_d1' % t] = [ch['hd'].reshape(N, -1).T @ gt, gt.sum(0)]
        gh = (gt @ W.T).reshape(ch['hd'].shape)
        gh = gh / 0.5 * np.asarray(masks['%s_l1drop' % t], np.float64)
        for i in range(5, 0, -1):
            rec = ch['convs'][i - 1]
            if i in (2, 4):
                gh = pool_bwd('%s_conv%d' % (t, i), rec['p'], rec['pooled'], gh)
            gh, da = act_bwd('%s_conv%d' % (t, i), rec['yb'], rec['r'], rec['alpha'], gh)
            grads['%s_ch_prelu%d' % (t, i)] = [da]
            dgamma = (gh * rec['xhat']).sum(axis=(0, 2, 3))
            dbeta = gh.sum(axis=(0, 2, 3))
            M = gh.shape[0] * gh.shape[2] * gh.shape[3]
            sh = (1, -1, 1, 1)
            dzc = (rec['gamma'] * rec['inv_std']).reshape(sh) / M * (
                M * gh - dbeta.reshape(sh) - rec['xhat'] * dgamma.reshape(sh))
            name = '%s_ch_conv%d_bn' % (t, i)
            grads[name] = [dbeta, dgamma, np.zeros_like(dbeta), np.zeros_like(dbeta)]
            gh, dW = _conv_bwd(rec['x'], rec['W'], dzc)
            grads['%s_ch_conv%d' % (t, i)] = [dW]
    if debug is not None:
        debug['forced'] = forced
    return loss, grads, bn_batch, probs


def adam_step(params, grads, state, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, trainable=None):
    """lasagne.updates.adam: t += 1; a_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= a_t*m/(sqrt(v)+eps).
    state: {'t': int, 'm': {name:[...]}, 'v': {...}} (created on first use). Returns new params (float64)."""
    if 't' not in state:
        state['t'] = 0
        state['m'] = {k: [np.zeros_like(np.asarray(a, np.float64)) for a in v] for k, v in params.items()}
        state['v'] = {k: [np.zeros_like(np.asarray(a, np.float64)) for a in v] for k, v in params.items()}
    state['t'] += 1
    t = state['t']
    a_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    new = {}
    for k, arrs in params.items():
        out = []
        for i, p in enumerate(arrs):
            p = np.asarray(p, np.float64)
            if trainable is not None and not trainable(k, i):
                out.append(p.copy())
                continue
            g = grads[k][i]
            m = state['m'][k][i] = b1 * state['m'][k][i] + (1 - b1) * g
            v = state['v'][k][i] = b2 * state['v'][k][i] + (1 - b2) * g * g
            out.append(p - a_t * m / (np.sqrt(v) + eps))
        new[k] = out
    return new


def bn_running_update(params, bn_batch, alpha=BN_ALPHA):
    """BatchNormLayer running averages: new = (1-alpha)*old + alpha*batch, for mean and inv_std."""
    new = {k: [np.asarray(a, np.float64).copy() for a in v] for k, v in params.items()}
    for name, (mean, inv_std) in bn_batch.items():
        new[name][2] = (1 - alpha) * new[name][2] + alpha * mean
        new[name][3] = (1 - alpha) * new[name][3] + alpha * inv_std
    return new
