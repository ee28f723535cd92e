"""
build_model() and the `net` object protocol of the reference, on libscs (HIP, gfx950).

Mirrors /root/reference/cnn_cort/nets.py:127-255.  The reference returns a nolearn ``NeuralNet`` wrapping a
Lasagne graph; both are external packages.  ``Net`` below reproduces the part of that object the reference
uses: fit / predict / predict_proba / load_params_from / save_params_to / train_history_ (call sites
nets.py:154-156,233-251, base.py:425,435, train_model.py:57-60), with the training-loop semantics of
nolearn 0.6.0 + nolearn_utils 0.2.2 restated here:

  * TrainSplit(eval_size): first fold of a non-shuffled stratified K-fold with K = round(1/eval_size) is the
    validation set (old sklearn.cross_validation.StratifiedKFold semantics, restated in `stratified_first_fold`).
  * BatchIterator: consecutive mini-batches, no shuffling.  The reference never passes a batch iterator, so
    nolearn's default of 128 applies and options['batch_size'] is ignored.  Here options['batch_size']
    is honoured when present (documented deviation); absent -> 128.
  * per epoch: train on all training batches (dropout on, BN batch statistics), then evaluate deterministically
    on the validation batches; history row {'epoch','train_loss','train_loss_best','valid_loss',
    'valid_loss_best','valid_accuracy','dur'} (schema pinned by nets/miccai2012_v1/*_history.pkl).
  * on_epoch_finished = [SaveWeights(only_best), SaveTrainingHistory, EarlyStopping(patience)] (nets.py:238-242);
    early stopping fires when epoch - best_epoch >= patience (pinned by the shipped history: best 42, stop 62)
    and restores the best weights.
  * optimiser Adam(lr 1e-3) on mean categorical cross-entropy (nets.py:235-237).

Deviations (documented in DESIGN.md): weights that fail to load raise instead of silently leaving a random
net (nets.py:249-253 swallows every exception); files are a float32 blob + JSON manifest instead of pickles
(the reference's own .pkl is still readable).
"""
import json
import os
import threading
import time
import warnings

import numpy as np

from . import params as P
from .load_options import is_true

__all__ = ['build_model', 'Net', 'stratified_first_fold', 'EarlyStopping']

_SUPER = 16384      # patches moved to the device per slice when the caller hands over host arrays


def stratified_first_fold(y, n_folds):
    """Validation indices = fold 0 of sklearn<=0.17's StratifiedKFold(y, n_folds) (no shuffle), which is what
    nolearn's TrainSplit takes: per class, the first ceil-share of that class's samples in order of appearance."""
    y = np.asarray(y)
    n = y.shape[0]
    uniq, y_inv = np.unique(y, return_inverse=True)
    test_folds = np.zeros(n, dtype=np.int64)
    per_label = []
    for c in np.bincount(y_inv):
        m = max(int(c), n_folds)
        sizes = np.full(n_folds, m // n_folds, dtype=np.int64)
        sizes[:m % n_folds] += 1
        bounds = np.concatenate([[0], np.cumsum(sizes)])
        per_label.append([np.arange(bounds[i], bounds[i + 1]) for i in range(n_folds)])
    for fold in range(n_folds):
        for li in range(len(uniq)):
            idx = np.nonzero(y_inv == li)[0]
            split = per_label[li][fold]
            split = split[split < idx.size]
            test_folds[idx[split]] = fold
    valid = np.nonzero(test_folds == 0)[0]
    train = np.nonzero(test_folds != 0)[0]
    return train, valid


class EarlyStopping:
    """nolearn_utils.hooks.EarlyStopping with the stop rule the shipped history pins (see module docstring)."""

    def __init__(self, patience=20):
        self.patience = patience
        self.best_valid = np.inf
        self.best_valid_epoch = 0
        self.best_weights = None

    def __call__(self, net, train_history):
        cur = train_history[-1]['valid_loss']
        epoch = train_history[-1]['epoch']
        if cur < self.best_valid:
            self.best_valid = cur
            self.best_valid_epoch = epoch
            self.best_weights = net.get_all_params_values()
        elif epoch - self.best_valid_epoch >= self.patience:
            if net.verbose:
                print('Early stopping.')
                print('Best valid loss was {:.6f} at epoch {}.'.format(self.best_valid, self.best_valid_epoch))
            net.load_params_from(self.best_weights)
            raise StopIteration()


def rotate_flip_views(views, indices, choices):
    """The on-the-fly augmentation of the reference's Rotate_batch_Iterator.transform (nets.py:41-124; defined there
    but never attached to the net, so it is opt-in here: options['augment']).  For the samples `indices` (half of the
    batch, the same for all three views) each view [B,1,32,32] is replaced by one of three variants, chosen per
    sample and per view by `choices[v][i]` in {0,1,2}: 0 = rotated by 180 degrees, 1 = flipped left-right,
    2 = rotated by 180 degrees then flipped left-right (= flipped up-down).  torch tensors on any device; returns new
    tensors, the inputs are not modified."""
    out = []
    for v, x in enumerate(views):
        x = x.clone()
        sel = x[indices]
        r = choices[v].to(sel.device).view(-1, 1, 1, 1)
        aug = sel.flip(2, 3)                                    # rotate_180
        aug = aug.where(r != 1, sel.flip(3))                    # rotate_0_flipped
        aug = aug.where(r != 2, sel.flip(2))                    # rotate_180_flipped
        x[indices] = aug
        out.append(x)
    return out


class Net:
    """Drop-in for the nolearn NeuralNet the reference builds at nets.py:233-246."""

    def __init__(self, weights_file=None, history_file=None, max_epochs=100, patience=20, eval_size=0.25,
                 batch_size=128, verbose=0, device=None, update_learning_rate=0.001, seed=0, augment=False,
                 fit_slice=16384, fit_resident_bytes=2 << 30, fwd_split=False, fit_split=False):
        self.augment = bool(augment)
        # inference (predict / predict_proba / test_scan's patch entries) on the split kernels: convolutions on the f16 matrix cores
        # with every f32 operand as two f16 pieces, f32-grade results at ~2x the rate (csrc/conv_split.h; option fwd_split of libscs)
        self.fwd_split = bool(fwd_split)
        # training: the forward pass of the conv1 + conv2 block on the split kernels (csrc/conv_split_train.h; option train_split of
        # libscs - named fit_split here: `train_split` is the reference's hold-out fraction, configuration.cfg [model])
        self.fit_split = bool(fit_split)
        self.fit_slice = int(fit_slice)                      # samples per streamed slice of a materialised training set
        self.fit_resident_bytes = int(fit_resident_bytes)    # sets up to this size are uploaded once instead
        self.weights_file = weights_file
        self.history_file = history_file
        self.max_epochs = int(max_epochs)
        self.patience = int(patience)
        self.eval_size = float(eval_size)
        self.batch_size = int(batch_size)
        self.verbose = int(verbose)
        self.update_learning_rate = float(update_learning_rate)
        self.seed = int(seed)
        self.device = device
        self.train_history_ = []
        self._params = None          # host copy, Lasagne layout (OrderedDict)
        self._handle = None          # libscs handle (created lazily on the device)
        self._handle_lock = threading.RLock()
        self._tls = threading.local()     # per-thread side streams (side_stream)
        self._dirty = True           # host params newer than device
        self._dev_newer = False      # device params newer than host (after training steps)
        self._step = 0
        self.gpu_seconds = 0.0       # device time of predict_volume calls (HIP events; bench.py --mode cohort reads it)
        self.last_scan_voxels = 0

    # ------------------------------------------------------------------ device plumbing
    def _torch(self):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('cnn_cort requires a HIP device (MI355X); torch.cuda.is_available() is False')
        return torch

    def _dev(self):
        if self.device is None:
            self.device = int(os.environ.get('LOCAL_RANK', '0'))
        return int(self.device)

    def _ensure_handle(self):
        """The libscs handle with the current parameters on the device.  Check-then-act on shared state: test_cohort's reader
        threads (normalize_volume, roi_centers) and the main thread may all arrive at a cold Net at once, so creation and the
        parameter upload happen under a lock - one handle, parameters set before anyone sees it."""
        from . import _hip
        torch = self._torch()
        with self._handle_lock:
            if self._handle is None:
                torch.cuda.set_device(self._dev())
                self._handle = _hip.Handle(self._dev())
                if self.fwd_split:
                    self._handle.set_option('fwd_split', 1)
                if self.fit_split:
                    self._handle.set_option('train_split', 1)
            if self._params is None:
                self._params = P.init_params(self.seed)       # Lasagne default initialisers (training from scratch)
                self._dirty = True
            if self._dirty:
                self._handle.set_params(P.params_to_flat(self._params))
                self._dirty = False
                self._dev_newer = False
            return self._handle

    def _sync_host(self):
        if self._dev_newer and self._handle is not None:
            self._params = P.flat_to_params(self._handle.get_params())
            self._dev_newer = False

    # ------------------------------------------------------------------ parameters
    def get_all_params_values(self):
        self._sync_host()
        if self._params is None:
            self._params = P.init_params(self.seed)
        return {k: [a.copy() for a in v] for k, v in self._params.items()}

    def load_params_from(self, source):
        """nolearn load_params_from: a path (our .f32/.json pair or a reference .pkl) or a dict of arrays."""
        if isinstance(source, (str, os.PathLike)):
            params = P.load_params(str(source))
        else:
            params = P.check_params({k: [np.asarray(a, np.float32) for a in v] for k, v in source.items()})
        self._params = params
        self._dirty = True
        self._dev_newer = False
        return self

    def save_params_to(self, path):
        self._sync_host()
        if self._params is None:
            self._params = P.init_params(self.seed)
        return P.save_params(str(path), self._params)

    # ------------------------------------------------------------------ inference
    def _as_device_inputs(self, X, lo, hi):
        torch = self._torch()
        out = []
        for key in ('in1', 'in2', 'in3', 'in4'):
            a = X[key]
            if isinstance(a, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(a[lo:hi], dtype=np.float32)).to('cuda:%d' % self._dev())
            else:
                t = a[lo:hi].to(device='cuda:%d' % self._dev(), dtype=torch.float32).contiguous()
            out.append(t)
        return out

    @staticmethod
    def _check_X(X):
        n = None
        for key, tail in (('in1', (1, 32, 32)), ('in2', (1, 32, 32)), ('in3', (1, 32, 32)), ('in4', (15,))):
            if key not in X:
                raise ValueError("input dict lacks '%s'" % key)
            shp = tuple(X[key].shape)
            if shp[1:] != tail:
                raise ValueError("'%s' has shape %r, expected [N,%s]" % (key, shp, ','.join(map(str, tail))))
            if n is None:
                n = shp[0]
            elif shp[0] != n:
                raise ValueError('inputs disagree on the number of samples')
        return n

    def _device_slices(self, X, n):
        """Yield (lo, hi, [ax, co, sa, atlas] on the device) in slices of _SUPER patches.  Host arrays are uploaded
        by a helper thread on its own stream, one slice ahead, so that the transfer of slice i+1 runs under the
        kernels of slice i (1.23 GB cross PCIe per 100 000 patches)."""
        torch = self._torch()
        bounds = [(lo, min(n, lo + _SUPER)) for lo in range(0, n, _SUPER)]
        if len(bounds) <= 1 or not any(isinstance(X[k], np.ndarray) for k in ('in1', 'in2', 'in3', 'in4')):
            for lo, hi in bounds:
                yield lo, hi, self._as_device_inputs(X, lo, hi)
            return
        import queue
        import threading
        dev = self._dev()
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        q = queue.Queue(maxsize=2)

        def uploader():
            try:
                with torch.cuda.stream(side):
                    for lo, hi in bounds:
                        ts = self._as_device_inputs(X, lo, hi)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        q.put((lo, hi, ts, ev))
                q.put(None)
            except BaseException as e:            # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=uploader, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                lo, hi, ts, ev = item
                main.wait_event(ev)
                for t in ts:
                    t.record_stream(main)
                yield lo, hi, ts
        finally:
            while th.is_alive():                  # drain so that the helper can finish if the consumer stopped early
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()

    def _forward(self, X, want_probs):
        torch = self._torch()
        n = self._check_X(X)
        h = self._ensure_handle()
        dev = 'cuda:%d' % self._dev()
        probs = np.empty((n, 15), np.float32) if want_probs else None
        if n and all(isinstance(X[k], np.ndarray) for k in ('in1', 'in2', 'in3', 'in4')):
            # host arrays (what the reference hands over): the library's own double-buffered host entry, which
            # uploads chunk i+1 under the kernels of chunk i
            host = [np.ascontiguousarray(X[k], dtype=np.float32) for k in ('in1', 'in2', 'in3', 'in4')]
            lab32 = np.empty((n,), np.int32)
            h.forward_host(host[0], host[1], host[2], host[3], n, probs, lab32)
            return probs, lab32.astype(np.int64)
        labels = np.empty((n,), np.int64)
        stream = torch.cuda.current_stream(self._dev()).cuda_stream
        pending = None                                  # outputs of the previous slice, copied back one slice late
        for lo, hi, (ax, co, sa, at) in self._device_slices(X, n):
            d_probs = torch.empty((hi - lo, 15), dtype=torch.float32, device=dev) if want_probs else None
            d_lab = torch.empty((hi - lo,), dtype=torch.int32, device=dev)
            h.forward(ax, co, sa, at, hi - lo, d_probs, d_lab, stream)
            if pending is not None:
                self._collect(pending, probs, labels)
            pending = (lo, hi, d_probs, d_lab)
        if pending is not None:
            self._collect(pending, probs, labels)
        return probs, labels

    @staticmethod
    def _collect(pending, probs, labels):
        lo, hi, d_probs, d_lab = pending
        if d_probs is not None:
            probs[lo:hi] = d_probs.cpu().numpy()
        labels[lo:hi] = d_lab.cpu().numpy()

    def predict_proba(self, X):
        """float32 [N,15] softmax outputs (deterministic forward) - base.py:435-438."""
        return self._forward(X, True)[0]

    def predict(self, X):
        """int64 [N] = argmax(predict_proba) - base.py:425-428 (nolearn, use_label_encoder=False)."""
        return self._forward(X, False)[1]

    def _wait(self, stream=None):
        """Sleep (not spin) until the work queued on `stream` (default: the current one) is done: the event is polled between
        short sleeps.  The runtime's own synchronisation - a .cpu() copy, hipEventSynchronize even with the blocking flag -
        busy-waits on this platform: a subject's 0.12 s GPU stage was 0.12 CPU-seconds of spinning."""
        torch = self._torch()
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream(self._dev()))
        delay = 1e-4
        while not ev.query():
            time.sleep(delay)
            delay = min(1e-3, delay * 1.5)

    _warned_split_shared = False      # (class-wide: the fwd_split / shared-towers note is printed once per process)
    SHARED_MIN_VOXELS = 4096          # below this the patch path is as fast
    SHARED_MAX_BOX_RATIO = 24         # bounding-box voxels per centre up to which sharing the tower work pays
    SHARED_MAX_VOXELS = 1 << 22       # 'auto' keeps 6.5 KB of block-5 features per centre on the device: 27 GB here

    def predict_volume(self, volume, centers, atlas4d=None, want_probs=False, batch=100000, atlas_rows=None, towers='auto'):
        """Gather-fused inference for test_scan: `volume` float32 [X,Y,Z] already z-scored, `centers` int [N,3], and
        the priors either as `atlas4d` float32 [X,Y,Z,15] (gathered on the device, background rule applied there)
        or as `atlas_rows` float32 [N,15] already gathered by the caller with the rule applied (base.atlas_vectors_at).
        `volume` and `centers` may be tensors already on the device (normalize_volume / roi_centers): nothing is uploaded then.
        towers: 'patch' = one tower evaluation per voxel and view (the reference's procedure); 'shared' = neighbouring voxels
        share the tower work (csrc/dense_eval.h: same kernels on tiles of per-slice maps, the same bits, ~5x less GPU time on
        an ROI); 'auto' = shared when the centres fill their bounding box densely enough and the priors come as rows.
        Returns (labels int64 [N], probs float32 [N,15] or None).
        Replaces load_patch_batch + net.predict/predict_proba (base.py:380-397, 425-438)."""
        if towers not in ('auto', 'patch', 'shared'):
            raise ValueError("towers must be 'auto', 'patch' or 'shared'")
        torch = self._torch()
        dev = 'cuda:%d' % self._dev()
        vol_on_dev = torch.is_tensor(volume)
        if vol_on_dev:
            if volume.dtype != torch.float32 or not volume.is_cuda or not volume.is_contiguous():
                raise ValueError('a device volume must be a contiguous float32 tensor')
        else:
            volume = np.ascontiguousarray(volume, dtype=np.float32)
        shape = tuple(int(d) for d in volume.shape)
        if len(shape) != 3:
            raise ValueError('volume must be 3-D, got shape %r' % (shape,))
        cen_on_dev = torch.is_tensor(centers)
        if cen_on_dev:
            if centers.dtype != torch.int32 or not centers.is_cuda or not centers.is_contiguous() or centers.dim() != 2 or centers.shape[1] != 3:
                raise ValueError('device centres must be a contiguous int32 [N,3] tensor')
        else:
            centers = np.ascontiguousarray(np.asarray(centers).reshape(-1, 3), dtype=np.int32)
        if (atlas4d is None) == (atlas_rows is None):
            raise ValueError('give exactly one of atlas4d and atlas_rows')
        n = int(centers.shape[0])
        if atlas_rows is not None:
            atlas_rows = np.ascontiguousarray(atlas_rows, dtype=np.float32)
            if atlas_rows.shape != (n, 15):
                raise ValueError('atlas_rows must be [%d, 15], got %r' % (n, atlas_rows.shape))
        else:
            atlas4d = np.ascontiguousarray(atlas4d, dtype=np.float32)
            if tuple(atlas4d.shape) != shape + (15,):
                raise ValueError('atlas4d shape %r does not match volume shape %r + (15,)' % (atlas4d.shape, shape))
        if n and not cen_on_dev and ((centers < 0).any() or (centers >= np.asarray(shape, np.int32)).any()):
            raise ValueError('patch centres outside the volume')          # (device lists come from roi_centers: inside by construction)
        if n == 0:
            return np.zeros((0,), np.int64), (np.zeros((0, 15), np.float32) if want_probs else None)
        h = self._ensure_handle()
        stream = torch.cuda.current_stream(self._dev()).cuda_stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        d_vol = volume if vol_on_dev else torch.from_numpy(volume).to(dev)
        d_atl = torch.from_numpy(atlas_rows if atlas_rows is not None else atlas4d).to(dev, non_blocking=True)
        d_cen = centers if cen_on_dev else torch.from_numpy(centers).to(dev)
        d_lab = torch.empty((n,), dtype=torch.int32, device=dev)
        d_pr = torch.empty((n, 15), dtype=torch.float32, device=dev) if want_probs else None
        batch = max(1, int(batch))
        box = None
        if towers != 'patch' and atlas_rows is not None:
            if cen_on_dev:
                box = torch.stack([d_cen.amin(0), d_cen.amax(0)], 1).flatten().cpu().numpy()
            else:
                box = np.stack([centers.min(0), centers.max(0)], 1).ravel()
            extent = (box[1::2] - box[0::2] + 1).astype(np.int64)
            if towers == 'auto' and (n < self.SHARED_MIN_VOXELS or n > self.SHARED_MAX_VOXELS or
                                     int(extent.prod()) > self.SHARED_MAX_BOX_RATIO * n):
                box = None
        elif towers == 'shared':
            raise ValueError("towers='shared' needs the priors as atlas_rows")
        self.last_scan_towers = 'shared' if box is not None else 'patch'
        if box is not None and self.fwd_split and not Net._warned_split_shared:
            # the shared-tower evaluation (csrc/dense_eval.h) runs the exact-f32 kernels on its own tile sizes; the split kernels
            # exist for the per-voxel entries only
            Net._warned_split_shared = True
            warnings.warn("fwd_split does not change towers='%s' on a dense ROI: the shared-tower evaluation stays on the f32 kernels "
                          "(and is the faster form for a whole subject); options['towers'] = 'patch' runs the per-voxel towers on the "
                          "split kernels" % towers, RuntimeWarning, stacklevel=2)
        if box is not None:
            h.forward_volume_shared(d_vol, shape, d_cen, d_atl, n, box, d_pr if want_probs else None, d_lab, stream)
        for lo in range(0, n if box is None else 0, batch):
            hi = min(n, lo + batch)
            if atlas_rows is not None:
                h.forward_volume_rows(d_vol, shape, d_cen[lo:hi], d_atl[lo:hi], hi - lo,
                                      d_pr[lo:hi] if want_probs else None, d_lab[lo:hi], stream)
            else:
                h.forward_volume(d_vol, shape, d_cen[lo:hi], d_atl, hi - lo,
                                 d_pr[lo:hi] if want_probs else None, d_lab[lo:hi], stream)
        ev1.record()
        self._wait()
        labels = d_lab.cpu().numpy().astype(np.int64)
        probs = d_pr.cpu().numpy() if want_probs else None
        self.gpu_seconds += ev0.elapsed_time(ev1) * 1e-3
        self.last_scan_voxels = n
        return labels, probs

    def side_stream(self):
        """Context manager that makes a stream of the calling THREAD's own the current one (test_cohort's reader threads prepare
        the next subjects on the device while the main thread's stream runs the classifier)."""
        torch = self._torch()
        if getattr(self._tls, 'stream', None) is None:
            self._tls.stream = torch.cuda.Stream(device=self._dev())
        return torch.cuda.stream(self._tls.stream)

    @staticmethod
    def _file_order(a):
        """(buffer in memory order, dims (X, Y, Z), is-Fortran) of a 3-D array without copying it when it is contiguous either
        way (nifti.load returns Fortran-ordered arrays, like nibabel)."""
        a = np.asarray(a)
        if a.ndim != 3:
            raise ValueError('volume must be 3-D, got shape %r' % (a.shape,))
        if a.dtype == np.bool_:
            a = a.view(np.uint8)
        if a.dtype.byteorder == '>' or a.dtype.name not in ('uint8', 'int16', 'uint16', 'int32', 'float32', 'float64'):
            a = a.astype(np.float64 if a.dtype.kind == 'f' else np.int32 if a.dtype.kind in 'iu' and a.dtype.itemsize <= 4 else np.float64)
        if a.flags['F_CONTIGUOUS'] and not a.flags['C_CONTIGUOUS']:
            return a.T, a.shape, True                       # a.T is the C-contiguous view of the same bytes
        return np.ascontiguousarray(a), a.shape, False

    def normalize_volume(self, raw):
        """(image - mean) / std over the non-zero voxels (population std, in double, one rounding to float32) of a raw T1
        volume, ON the device: float32 tensor [X,Y,Z] (C order) ready for predict_volume.  The raw voxels go up as they lie in
        the file.  Replaces the whole-volume NumPy passes of base.py:357-358."""
        torch = self._torch()
        buf, dims, fortran = self._file_order(raw)
        h = self._ensure_handle()
        dev = 'cuda:%d' % self._dev()
        d_raw = torch.from_numpy(buf).to(dev)
        out = torch.empty(tuple(int(d) for d in dims), dtype=torch.float32, device=dev)
        scratch = torch.empty((h.STAT_DOUBLES,), dtype=torch.float64, device=dev)
        h.zscore_volume(d_raw, dims, fortran, out, scratch, torch.cuda.current_stream(self._dev()).cuda_stream)
        return out

    def roi_centers(self, mask, iterations=10):
        """The voxel list test_scan classifies (base.py:369-370): mask != 0, dilated `iterations` times (6-neighbour cross), its
        voxels in C order - on the device.  Returns (centers int64 ndarray [N,3], the same as an int32 device tensor)."""
        torch = self._torch()
        buf, dims, fortran = self._file_order(mask)
        h = self._ensure_handle()
        dev = 'cuda:%d' % self._dev()
        st = torch.cuda.current_stream(self._dev()).cuda_stream
        nvox = int(np.prod(dims))
        d_raw = torch.from_numpy(buf).to(dev)
        d_mask = torch.empty((nvox,), dtype=torch.uint8, device=dev)
        d_scr = torch.empty((nvox,), dtype=torch.uint8, device=dev)
        d_off = torch.empty(((nvox + 1023) // 1024 + 1,), dtype=torch.int32, device=dev)
        n = h.roi_prepare(d_raw, dims, fortran, iterations, d_mask, d_scr, d_off, st)
        d_cen = torch.empty((n, 3), dtype=torch.int32, device=dev)
        h.roi_centers(d_mask, dims, d_off, d_cen, n, st)
        self._wait()
        return d_cen.cpu().numpy().astype(np.int64), d_cen

    def dilate_mask(self, mask, iterations=10):
        """scipy.ndimage.binary_dilation(mask, iterations=iterations) (default 6-neighbour structure, border 0) on the GPU:
        bool ndarray of mask's shape.  test_scan grows the registered sub-cortical mask into its ROI with it (base.py:369)."""
        torch = self._torch()
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if m.ndim != 3:
            raise ValueError('mask must be 3-D, got shape %r' % (m.shape,))
        h = self._ensure_handle()
        dev = 'cuda:%d' % self._dev()
        d = torch.from_numpy(m).to(dev)
        scratch = torch.empty_like(d)
        h.dilate_mask(d, scratch, m.shape, iterations, torch.cuda.current_stream(self._dev()).cuda_stream)
        return d.cpu().numpy().astype(bool)

    # ------------------------------------------------------------------ training
    def _uploads_ahead(self, n_slices, fill):
        """Slices of host data to the device, one slice ahead of the consumer: a helper thread calls `fill(i, alloc)` for
        slice i - which gathers its rows straight into page-locked staging arrays obtained from `alloc(key, shape, dtype)`
        (two rotating sets, grown on demand) and returns the keys in consumer order - and copies them to fresh device
        tensors on its own stream, so that gather
        and transfer of slice i+1 run under the kernels of slice i.  Yields lists of device tensors that the consumer's
        current stream may use."""
        import queue
        import threading
        torch = self._torch()
        dev = self._dev()
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        q = queue.Queue(maxsize=1)                 # one slice in flight + one being consumed = two resident slices
        slots = [{'bufs': {}, 'free': None} for _ in range(2)]

        def uploader():
            try:
                with torch.cuda.stream(side):
                    for i in range(n_slices):
                        slot = slots[i % 2]
                        if slot['free'] is not None:
                            slot['free'].synchronize()       # the copy that last read this staging set has finished
                        used = {}

                        def alloc(k, shape, dtype):
                            need = int(np.prod(shape))
                            buf = slot['bufs'].get(k)
                            if buf is None or buf.numel() < need or buf.dtype != dtype:
                                buf = torch.empty((need,), dtype=dtype).pin_memory()
                                slot['bufs'][k] = buf
                            view = buf[:need].view(*shape)
                            used[k] = view
                            return view.numpy()

                        order = fill(i, alloc)               # the keys in the order the consumer expects (the gathers
                        views = [used[k] for k in order]     # themselves may have run in any order, in parallel)
                        ts = [torch.empty(v.shape, dtype=v.dtype, device='cuda:%d' % dev) for v in views]
                        for t, v in zip(ts, views):
                            t.copy_(v, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        slot['free'] = ev
                        q.put((ts, ev))
                q.put(None)
            except BaseException as e:            # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=uploader, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                ts, ev = item
                main.wait_event(ev)
                for t in ts:
                    t.record_stream(main)
                yield ts
        finally:
            while th.is_alive():                  # drain so that the helper can finish if the consumer stopped early
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()

    def fit(self, X, y=None, epochs=None):
        """net.fit({'in1','in2','in3','in4'}, y) - train_model.py:57-60.

        X may also be a cnn_cort.base.CompactTrainingSet (volumes + voxel rows + index permutation, from
        generate_training_set_compact): the patches are then cut inside the first conv kernel and never exist in memory.

        Materialised sets stay on the HOST; the device holds two slices of `self.fit_slice` samples at a time (default
        16 384: 2 x 202 MB), uploaded one slice ahead of the kernels, unless the whole set fits `self.fit_resident_bytes`
        (default 2 GiB), in which case it is uploaded once.

        Data parallel (torch.distributed initialised): mini-batch s*world + r goes to rank r, so one optimiser step sees
        world x batch_size samples at the unchanged learning rate (the reference is single-device; scale batch_size down by
        the world size for history-comparable runs); the hold-out pass is sharded over the ranks."""
        from . import dist
        from .base import CompactTrainingSet, compact_from_views
        torch = self._torch()
        if isinstance(X, dict) and y is not None:
            # the unchanged train_model.py sequence: the dict holds the lazy views generate_training_set returned -> the same
            # set in its compact form (with the prior rows and labels of THIS call), patches cut on the GPU
            lazy = compact_from_views(X, y)
            if lazy is not None:
                X = lazy
        compact = isinstance(X, CompactTrainingSet)
        if not compact:
            # anything array-like that is neither a NumPy array nor a torch tensor (a lone lazy view, a list) becomes the array
            X = {k: (v if isinstance(v, np.ndarray) or torch.is_tensor(v) else np.asarray(v)) for k, v in X.items()}
        if compact:
            n = len(X)
            y_all = X.y[X.index]                       # labels in training order
        else:
            n = self._check_X(X)
            if y is None:
                raise ValueError('fit() needs the labels y')
            y_all = np.asarray(y)
            if y_all.shape[0] != n:
                raise ValueError('y has %d labels for %d samples' % (y_all.shape[0], n))
        if y_all.size and (y_all.min() < 0 or y_all.max() > 14):
            raise ValueError('labels must lie in 0..14')
        h = self._ensure_handle()
        dev = 'cuda:%d' % self._dev()
        stream = torch.cuda.current_stream(self._dev()).cuda_stream
        world, rank = dist.world_size(), dist.rank()

        if self.eval_size and self.eval_size > 0:
            tr_idx, va_idx = stratified_first_fold(y_all, int(round(1.0 / self.eval_size)))
        else:
            tr_idx, va_idx = np.arange(n), np.arange(0)
        ntr, nva = len(tr_idx), len(va_idx)
        bs = self.batch_size
        starts = list(range(0, ntr, bs))
        my_starts = starts[rank::world]                # lock-step schedule: step s gives rank r the batch s*world + r
        n_steps = (len(starts) + world - 1) // world
        va_lo, va_hi = dist.shard_range(nva)           # this rank's share of the hold-out pass
        keys = ('in1', 'in2', 'in3', 'in4')

        if compact:
            offs, dims = X.descriptors()
            d_vols = torch.from_numpy(np.concatenate([v.ravel() for v in X.volumes])).to(dev)
            desc = np.zeros((len(offs), 3), np.int64)          # scs_volume: int64 offset, int32 x, y, z, reserved
            desc_i32 = desc.view(np.int32).reshape(len(offs), 6)
            for i, (o, d) in enumerate(zip(offs, dims)):
                desc[i, 0] = o
                desc_i32[i, 2:5] = d
            d_desc = torch.from_numpy(desc).to(dev)
            d_samples = torch.from_numpy(X.samples).to(dev)
            d_atlas = torch.from_numpy(X.atlas).to(dev)
            # the labels the caller passed (and that were range-checked and split above) are the labels trained on: X.y, not a
            # second reading of the label volumes (a set built with relabelled / merged classes must train on ITS labels)
            if X.y.size and int(X.y.max()) > 14:
                raise ValueError('labels must lie in 0..14')
            d_y = torch.from_numpy(X.y).to(dev)
            d_tr = torch.from_numpy(X.index[tr_idx].astype(np.int32)).to(dev)
            d_va = torch.from_numpy(X.index[va_idx].astype(np.int32)).to(dev)

            def compact_batch(seed, lo, hi):
                flips = None
                if self.augment and hi - lo > 1:
                    # the same draws as _train_batch makes for materialised patches; variant r of Rotate_batch_Iterator as
                    # mirror flags of the gather: 0 rotate_180 = rows + columns, 1 flipped = columns, 2 rotate_180 flipped = rows
                    nb = hi - lo
                    aug_rng.manual_seed((self.seed << 20) + self._step * 8 + rank)
                    idx = torch.randperm(nb, generator=aug_rng, device=dev)[:nb // 2]
                    code = torch.tensor([3, 2, 1], dtype=torch.uint8, device=dev)
                    flips = torch.zeros((nb,), dtype=torch.uint8, device=dev)
                    for v in range(3):
                        r = torch.randint(0, 3, (idx.numel(),), generator=aug_rng, device=dev)
                        flips[idx] |= code[r] << (2 * v)
                h.train_step_volume(d_vols, d_desc, d_samples, d_atlas, d_y, d_tr[lo:hi], hi - lo, seed=seed, want_loss=False,
                                    stream=stream, flips=flips)

            def train_batches():
                for lo in my_starts:
                    hi = min(ntr, lo + bs)
                    yield hi - lo, (lambda seed, lo=lo, hi=hi: compact_batch(seed, lo, hi))

            def evaluate():
                return h.eval_batch_volume(d_vols, d_desc, d_samples, d_atlas, d_y, d_va[va_lo:va_hi], va_hi - va_lo, stream=stream)
        else:
            per_sample = 3 * 4096 + 60 + 1
            on_device = all((not isinstance(X[k], np.ndarray)) and X[k].is_cuda for k in keys)     # caller's tensors already in HBM
            resident = on_device or (ntr + nva) * per_sample <= self.fit_resident_bytes
            sl = max(bs, (self.fit_slice // bs) * bs)              # slice = whole mini-batches

            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=4)           # the four row gathers of a slice run side by side (np.take drops the GIL)
            dts = {'in1': torch.float32, 'in2': torch.float32, 'in3': torch.float32, 'in4': torch.float32}

            def host_rows(idx, alloc=None):
                """Rows `idx` of the four inputs and the labels; into page-locked staging arrays when `alloc` is given."""
                def rows(k):
                    a = X[k]
                    if not isinstance(a, np.ndarray):
                        a = a[torch.as_tensor(idx)].cpu().numpy()
                        idx_k = None
                    else:
                        idx_k = idx
                    shape = (len(idx),) + tuple(a.shape[1:])
                    out = alloc(k, shape, dts[k]) if alloc else np.empty(shape, np.float32)
                    if idx_k is None:
                        out[...] = a
                    elif a.dtype == np.float32:
                        np.take(a, idx_k, axis=0, out=out)
                    else:
                        out[...] = np.take(a, idx_k, axis=0)
                    return out
                arrs = list(pool.map(rows, keys))
                yo = alloc('y', (len(idx),), torch.uint8) if alloc else np.empty((len(idx),), np.uint8)
                yo[...] = y_all[idx]
                return list(keys) + ['y'] if alloc else arrs + [yo]

            my_rows = np.concatenate([tr_idx[lo:min(ntr, lo + bs)] for lo in my_starts]) if my_starts else np.arange(0)
            my_sizes = [min(ntr, lo + bs) - lo for lo in my_starts]
            def device_rows(idx):
                if not on_device:
                    return [torch.from_numpy(a).to(dev) for a in host_rows(idx)]
                it = torch.as_tensor(idx, device=dev)
                return [X[k].to(dev)[it].to(torch.float32).contiguous() for k in keys] + \
                       [torch.from_numpy(np.ascontiguousarray(y_all[idx], dtype=np.uint8)).to(dev)]

            if resident:
                r_tr, r_va = device_rows(my_rows), device_rows(va_idx[va_lo:va_hi])

            def train_slices():
                if resident:
                    yield r_tr, my_sizes
                    return
                per_slice = max(1, sl // bs)                        # my mini-batches per slice
                groups = [my_sizes[i:i + per_slice] for i in range(0, len(my_sizes), per_slice)]
                bounds = np.concatenate([[0], np.cumsum([sum(g) for g in groups])]).astype(np.int64)
                fill = lambda i, alloc: host_rows(my_rows[bounds[i]:bounds[i + 1]], alloc)
                for ts, g in zip(self._uploads_ahead(len(groups), fill), groups):
                    yield ts, g

            def train_batches():
                for ts, sizes in train_slices():
                    at = 0
                    for nb in sizes:
                        views = [ts[0][at:at + nb], ts[1][at:at + nb], ts[2][at:at + nb]]
                        atl, yy = ts[3][at:at + nb], ts[4][at:at + nb]
                        at += nb
                        yield nb, (lambda seed, views=views, atl=atl, yy=yy, nb=nb: self._train_batch(h, views, atl, yy, nb, seed, stream, aug_rng, rank))

            def evaluate():
                if resident:
                    return h.eval_batch(r_va[0], r_va[1], r_va[2], r_va[3], r_va[4], va_hi - va_lo, stream=stream)
                ls, nc = 0.0, 0
                mine = va_idx[va_lo:va_hi]
                los = list(range(0, len(mine), sl))
                for ts in self._uploads_ahead(len(los), lambda i, alloc: host_rows(mine[los[i]:los[i] + sl], alloc)):
                    a, b = h.eval_batch(ts[0], ts[1], ts[2], ts[3], ts[4], int(ts[4].shape[0]), stream=stream)
                    ls += a
                    nc += b
                return ls, nc

        hooks = [self._hook_save_weights, self._hook_save_history, EarlyStopping(self.patience)]
        aug_rng = torch.Generator(device=dev) if self.augment else None
        best_train, best_valid = np.inf, np.inf
        first_epoch = len(self.train_history_)
        if self.verbose:
            print('  epoch    trn loss    val loss    trn/val    valid acc  dur')
            print('-------  ----------  ----------  ---------  -----------  -----')
        # one rank: nothing sits between a step's gradients and their update, so the step applies lasagne.updates.adam itself,
        # group by group as the gradients become final (scs_set_fused_adam: same arithmetic, no launch behind the step)
        fused = not dist.is_dist()
        h.set_fused_adam(fused, lr=self.update_learning_rate)
        try:
            for ep in range(epochs or self.max_epochs):
                t0 = time.time()
                # batch losses are collected on the device and read back once per epoch (no per-step synchronisation)
                loss_buf = torch.zeros((max(1, n_steps),), dtype=torch.float32, device=dev)
                sizes, loss_view = [], None
                batches = train_batches()
                for step in range(n_steps):
                    item = next(batches, None)
                    active = item is not None
                    if active:
                        nb, run = item
                        self._step += 1
                        run((self.seed << 32) + self._step * 131 + rank)
                        if loss_view is None:
                            loss_view = dist.wrap_device_floats(h.train_loss_ptr(), 1, self._dev())
                        loss_buf[len(sizes):len(sizes) + 1].copy_(loss_view, non_blocking=True)
                        sizes.append(nb)
                    if not fused:
                        scale = dist.allreduce_grads(h, active, stream, n_active=dist.active_ranks(len(starts), step, world))
                        h.adam_apply(lr=self.update_learning_rate, grad_scale=scale, stream=stream)
                losses = loss_buf[:len(sizes)].cpu().numpy().astype(np.float64)
                self._dev_newer = True
                dist.sync_params(h)            # reconcile per-rank BN running statistics (no-op on one GPU)
                # nolearn: train_loss / valid_loss = batch losses averaged with the batch sizes as weights (a ragged last
                # batch counts for less); the hold-out pass sums per-sample losses, which is the same mean
                w = np.asarray(sizes, np.float64)
                loss_sum, cnt = dist.sum_pair(float((losses * w).sum()), float(w.sum()))
                train_loss = loss_sum / cnt if cnt else 0.0
                if self.fit_split and not np.isfinite(train_loss):
                    # option train_split of libscs: a conv1 activation beyond the f16 range of the split pieces turns that step's loss
                    # into NaN - and its gradients, which the (fused) Adam update has already applied: the weights are void from that
                    # step on (include/scs.h).  The reduced loss is the same number on every rank: all of them stop here.
                    raise FloatingPointError('fit_split: a scaled activation left the f16 range in epoch %d (NaN training loss); the weights are '
                                             'invalid from that step on - reload them and train with fit_split off (the exact-f32 step)'
                                             % (first_epoch + ep + 1))
                vloss, vcorrect = evaluate() if va_hi > va_lo else (0.0, 0)
                vloss, vcorrect = dist.sum_pair(vloss, float(vcorrect))
                valid_loss = vloss / nva if nva else 0.0
                valid_acc = vcorrect / nva if nva else 0.0
                best_train = min(best_train, train_loss)
                best_valid = min(best_valid, valid_loss)
                info = {'epoch': first_epoch + ep + 1, 'train_loss': train_loss, 'train_loss_best': best_train == train_loss,
                        'valid_loss': valid_loss, 'valid_loss_best': best_valid == valid_loss,
                        'valid_accuracy': valid_acc, 'dur': time.time() - t0}
                self.train_history_.append(info)
                if self.verbose:
                    print('{:>7}  {:>10.5f}  {:>10.5f}  {:>9.5f}  {:>11.5f}  {:.2f}s'.format(
                        info['epoch'], train_loss, valid_loss, train_loss / valid_loss if valid_loss else 0.0,
                        valid_acc, info['dur']))
                for hook in hooks:
                    hook(self, self.train_history_)
        except StopIteration:
            pass
        finally:
            h.set_fused_adam(False)            # the handle goes back to the two-call form (scs_train_step, scs_adam_apply)
            if not compact:
                pool.shutdown(wait=False)
        return self

    def _train_batch(self, h, views, atl, yy, nb, seed, stream, aug_rng, rank):
        if self.augment and nb > 1:
            torch = self._torch()
            dev = views[0].device
            aug_rng.manual_seed((self.seed << 20) + self._step * 8 + rank)
            idx = torch.randperm(nb, generator=aug_rng, device=dev)[:nb // 2]
            views = rotate_flip_views(views, idx, [torch.randint(0, 3, (idx.numel(),), generator=aug_rng, device=dev) for _ in range(3)])
        h.train_step(views[0], views[1], views[2], atl, yy, nb, seed=seed, want_loss=False, stream=stream)

    # hooks (nets.py:154-156, 238-242)
    def _hook_save_weights(self, net, history):
        from . import dist
        if self.weights_file and history[-1]['valid_loss_best'] and dist.rank() == 0:
            self.save_params_to(self.weights_file)

    def _hook_save_history(self, net, history):
        from . import dist
        if self.history_file and dist.rank() == 0:
            base = self.history_file[:-4] if self.history_file.endswith('.pkl') else self.history_file
            rows = [{k: (bool(v) if isinstance(v, (bool, np.bool_)) else v) for k, v in r.items()} for r in history]
            with open(base + '.json', 'w') as f:
                json.dump(rows, f, indent=0)
            if self.history_file.endswith('.pkl'):
                # what SaveTrainingHistory leaves (nets.py:156): a protocol-2 pickle of the list of epoch dicts, plain python
                # values, readable by the reference's python 2
                import pickle
                with open(self.history_file + '.tmp', 'wb') as f:
                    pickle.dump(rows, f, protocol=2)
                os.replace(self.history_file + '.tmp', self.history_file)


def build_model(weights_path, options):
    """
    Build the CNN model and return the net object (nets.py:127-255).
      weights_path: folder that holds <experiment>/<experiment>.{f32,json} (or the reference's .pkl)
      options: experiment, net_verbose, train_split, max_epochs, patience, patch_size, load_weights
               (+ batch_size, device - see module docstring)
    """
    name = options['experiment']
    folder = os.path.join(weights_path, name)
    os.makedirs(folder, exist_ok=True)                       # nets.py:140-143
    net_weights = os.path.join(folder, name + '.pkl')        # same stem as the reference; extension is replaced
    net_history = os.path.join(folder, name + '_history.pkl')
    ps = options['patch_size'][0] if isinstance(options['patch_size'], (list, tuple)) else int(options['patch_size'])
    if ps != 32:
        raise ValueError('the tri-planar net is built for 32x32 patches (patch_size=%r)' % (options['patch_size'],))
    net = Net(weights_file=net_weights, history_file=net_history,
              max_epochs=options['max_epochs'], patience=options['patience'], eval_size=options['train_split'],
              batch_size=options.get('batch_size') or 128, verbose=options['net_verbose'],
              device=options.get('device'), augment=is_true(options.get('augment')),
              seed=int(options.get('seed') or 0), fwd_split=is_true(options.get('fwd_split')), fit_split=is_true(options.get('fit_split')))
    if is_true(options['load_weights']):
        if net.verbose:
            print("    --> loading weights from ", net_weights)
        net.load_params_from(net_weights)                    # raises IOError / ValueError (reference: silent)
    return net
