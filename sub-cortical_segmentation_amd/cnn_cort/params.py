"""
Parameter store of the tri-planar net: layer inventory, file formats, initialisation.

The reference keeps parameters the way nolearn's ``save_params_to`` writes them: an
``OrderedDict{layer_name: [ndarray, ...]}`` over all 88 layers of the Lasagne graph built at
/root/reference/cnn_cort/nets.py:170-231 (56 of them carry arrays), pickled (nets.py:146,155,251).
This module keeps exactly that *logical* schema (names, order, per-layer array order, shapes,
float32) but stores it in a pickle-free container:

    <name>.f32   little-endian float32, all arrays back to back in layer order
    <name>.json  manifest: [{"name", "arrays": [{"shape", "offset"}]}...] (+ "format", "total")

``load_params`` additionally reads the reference's own ``.pkl`` through an allow-listed unpickler
(the pickle references only OrderedDict / numpy reconstruct helpers - pure data), so a user of the
reference can point ``net.load_params_from`` at the weights file they already have; ``save_params`` on a
``.pkl`` path additionally writes that pickle (python-2 cPickle protocol 2, numpy 1.x spelling), so the
reference can read weights trained here.

Array order inside a layer follows Lasagne's ``get_params()``:
  Conv2DLayer (bias stripped by batch_norm(), nets.py:171)  -> [W (Cout,Cin,3,3)]
  BatchNormLayer                                            -> [beta, gamma, mean, inv_std]
  ParametricRectifierLayer (prelu)                          -> [alpha]
  DenseLayer                                                -> [W (in,out), b]
"""
import json
import os
import pickle
from collections import OrderedDict

import numpy as np

TOWERS = ('axial', 'coronal', 'saggital')     # sic: 'saggital' is the key spelling (nets.py:202-212)
TOWER_INPUTS = ('in1', 'in2', 'in3')
CONV_CH = ((1, 20), (20, 20), (20, 40), (40, 40), (40, 60))   # (Cin, Cout) of conv1..5, nets.py:171-177
N_CLASSES = 15
N_ATLAS = 15
FC_CONV = 180                                   # nets.py:163
FC1_UNITS = 540                                 # nets.py:217
FC2_UNITS = 270                                 # nets.py:227


def layer_spec():
    """Ordered [(layer_name, [shape, ...])] for all 88 layers, as nolearn enumerates them."""
    spec = []
    for tower, inp in zip(TOWERS, TOWER_INPUTS):
        spec.append((inp, []))
        for i, (cin, cout) in enumerate(CONV_CH, start=1):
            spec.append(('%s_ch_conv%d' % (tower, i), [(cout, cin, 3, 3)]))
            spec.append(('%s_ch_conv%d_bn' % (tower, i), [(cout,)] * 4))
            spec.append(('%s_ch_conv%d_bn_nonlin' % (tower, i), []))
            spec.append(('%s_ch_prelu%d' % (tower, i), [(cout,)]))
            if i == 2:
                spec.append(('%s_max_pool_1' % tower, []))
            if i == 4:
                spec.append(('%s_max_pool_2' % tower, []))
        spec.append(('%s_l1drop' % tower, []))
        spec.append(('%s_d1' % tower, [(540, FC_CONV), (FC_CONV,)]))
        spec.append(('%s_prelu_d1' % tower, [(FC_CONV,)]))
    spec += [
        ('elem_channels', []), ('f1_drop', []),
        ('FC1', [(3 * FC_CONV, FC1_UNITS), (FC1_UNITS,)]), ('prelu_f1', [(FC1_UNITS,)]),
        ('f2_drop', []), ('in4', []), ('elem_channels2', []),
        ('fc_2', [(FC1_UNITS + N_ATLAS, FC2_UNITS), (FC2_UNITS,)]), ('prelu_f2', [(FC2_UNITS,)]),
        ('out_layer', [(FC2_UNITS, N_CLASSES), (N_CLASSES,)]),
    ]
    return spec


LAYER_SPEC = layer_spec()
N_STORED = sum(int(np.prod(s)) for _, shapes in LAYER_SPEC for s in shapes)   # 883 455


def is_trainable(layer_name, array_index):
    """BN mean / inv_std (arrays 2, 3 of a *_bn layer) are running statistics, not trainable."""
    return not (layer_name.endswith('_bn') and array_index >= 2)


def check_params(params):
    """Raise ValueError unless `params` matches LAYER_SPEC exactly (the reference silently random-
    initialises on any load failure, nets.py:249-253; we refuse instead - documented deviation)."""
    for name, shapes in LAYER_SPEC:
        if name not in params:
            raise ValueError('parameter file lacks layer %r' % name)
        arrs = params[name]
        if len(arrs) != len(shapes):
            raise ValueError('layer %r: %d arrays, expected %d' % (name, len(arrs), len(shapes)))
        for a, s in zip(arrs, shapes):
            if tuple(a.shape) != tuple(s):
                raise ValueError('layer %r: shape %r, expected %r' % (name, tuple(a.shape), s))
    return params


# ---------------------------------------------------------------------------------------------
# file formats
# ---------------------------------------------------------------------------------------------
class _DataOnlyUnpickler(pickle.Unpickler):
    """Unpickler that can only build OrderedDict / list / numpy arrays & scalars."""
    _ALLOWED = {
        ('collections', 'OrderedDict'),
        ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
        ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'),
        ('_codecs', 'encode'),            # how python-3 writers spell array bytes in protocol 2 (harmless)
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError('refusing to unpickle %s.%s (data-only loader)' % (module, name))


def read_reference_pickle(path):
    """Read a nolearn ``save_params_to`` / SaveTrainingHistory pickle (python-2 protocol 2)."""
    with open(path, 'rb') as f:
        return _DataOnlyUnpickler(f, encoding='latin1').load()


class _Py2PickleWriter:
    """Emits the exact opcode stream python-2 cPickle (protocol 2, what nolearn's ``save_params_to`` uses:
    ``pickle.dump(params, f, -1)``) produces for ``OrderedDict{str: [float32 ndarray, ...]}`` under numpy 1.x, so that the
    reference's own environment (python 2.7 + numpy 1.13, requirements.txt) can read files written here:
    ``numpy.core.multiarray._reconstruct`` (not numpy 2's ``numpy._core``), array bytes as a py2 ``str`` (BINSTRING), and
    cPickle's memo discipline (objects with a second reference are memoised: key strings, lists, the ndarray class, the
    shared '<f4' dtype, every array; temporaries - shape tuples, the raw byte strings - are not).  Given the arrays of
    nets/miccai2012_v1/miccai2012_v1.pkl it reproduces that file byte for byte (tests/test_params.py)."""

    def __init__(self):
        self.out = []
        self.memo_next = 1
        self.memo = {}                  # tag -> memo index of objects written before

    def _put(self, tag=None):
        i = self.memo_next
        self.memo_next += 1
        self.out.append(b'q' + bytes([i]) if i < 256 else b'r' + i.to_bytes(4, 'little'))
        if tag is not None:
            self.memo[tag] = i

    def _get(self, tag):
        i = self.memo[tag]
        self.out.append(b'h' + bytes([i]) if i < 256 else b'j' + i.to_bytes(4, 'little'))

    def _int(self, v):
        if 0 <= v < 256:
            self.out.append(b'K' + bytes([v]))
        elif 0 <= v < 65536:
            self.out.append(b'M' + v.to_bytes(2, 'little'))
        else:
            self.out.append(b'J' + int(v).to_bytes(4, 'little', signed=True))

    def _str(self, b, memoize):
        b = b if isinstance(b, bytes) else b.encode('latin1')
        self.out.append((b'U' + bytes([len(b)]) if len(b) < 256 else b'T' + len(b).to_bytes(4, 'little')) + b)
        if memoize:
            self._put()

    def _global(self, tag, module, name):
        if tag in self.memo:
            self._get(tag)
        else:
            self.out.append(b'c' + module.encode() + b'\n' + name.encode() + b'\n')
            self._put(tag)

    def _tuple(self, ints):
        if len(ints) > 3:
            self.out.append(b'(')
        for v in ints:
            self._int(int(v))
        self.out.append({0: b')', 1: b'\x85', 2: b'\x86', 3: b'\x87'}.get(len(ints), b't'))

    def _array(self, a):
        a = np.ascontiguousarray(a, dtype='<f4')
        self._global('reconstruct', 'numpy.core.multiarray', '_reconstruct')
        self._global('ndarray', 'numpy', 'ndarray')
        self._tuple((0,))
        self._str('b', False)
        self.out.append(b'\x87R')                      # (ndarray, (0,), 'b') -> _reconstruct(...)
        self._put()
        self.out.append(b'(')
        self._int(1)                                   # ndarray.__setstate__ version
        self._tuple(a.shape)
        if 'f4' in self.memo:
            self._get('f4')
        else:
            self._global('dtype', 'numpy', 'dtype')
            self._str('f4', False)
            self._int(0)
            self._int(1)
            self.out.append(b'\x87R')                  # dtype('f4', 0, 1)
            self._put('f4')
            self.out.append(b'(')
            self._int(3)
            self._str('<', False)
            self.out.append(b'NNN')
            self._int(-1)
            self._int(-1)
            self._int(0)
            self.out.append(b'tb')                     # dtype.__setstate__((3, '<', None, None, None, -1, -1, 0))
        self.out.append(b'\x89')                       # is_fortran = False
        self._str(a.tobytes(), False)
        self.out.append(b'tb')

    def _list_items(self, n, emit):
        """cPickle's batch_list for n < 1000 items: nothing / item APPEND / MARK items APPENDS."""
        if n == 1:
            emit(0)
            self.out.append(b'a')
        elif n > 1:
            self.out.append(b'(')
            for i in range(n):
                emit(i)
            self.out.append(b'e')

    def dump(self, params):
        items = list(params.items())
        assert len(items) < 1000
        self.out.append(b'\x80\x02')
        self._global('OrderedDict', 'collections', 'OrderedDict')
        self.out.append(b']')
        self._put()

        def pair(i):
            name, arrs = items[i]
            self.out.append(b']')
            self._put()

            def kv(j):
                if j == 0:
                    self._str(name, True)
                else:
                    self.out.append(b']')
                    self._put()
                    self._list_items(len(arrs), lambda k: self._array(arrs[k]))
            self._list_items(2, kv)
        self._list_items(len(items), pair)
        self.out.append(b'\x85R')                      # OrderedDict([[k, v], ...])
        self._put()
        self.out.append(b'.')
        return b''.join(self.out)


def write_reference_pickle(path, params):
    """Write `params` as the pickle nolearn's ``save_params_to`` writes under the reference's python 2 / numpy 1.x
    (nets.py:155 SaveWeights, :251 load_params_from): the reference can load weights trained here."""
    check_params(params)
    ordered = OrderedDict((name, list(params[name])) for name, _ in LAYER_SPEC)
    data = _Py2PickleWriter().dump(ordered)
    with open(path + '.tmp', 'wb') as f:
        f.write(data)
    os.replace(path + '.tmp', path)
    return path


def _paths(path):
    base = path[:-4] if path.endswith(('.f32', '.pkl')) else (path[:-5] if path.endswith('.json') else path)
    return base + '.f32', base + '.json'


def save_params(path, params):
    """Write the blob + manifest pair next to `path` (extension replaced)."""
    check_params(params)
    blob_path, man_path = _paths(path)
    manifest, off, chunks = [], 0, []
    for name, _ in LAYER_SPEC:
        entry = {'name': name, 'arrays': []}
        for a in params[name]:
            a = np.ascontiguousarray(a, dtype='<f4')
            entry['arrays'].append({'shape': list(a.shape), 'offset': off})
            chunks.append(a.ravel())
            off += a.size
        manifest.append(entry)
    tmp = blob_path + '.tmp'
    with open(tmp, 'wb') as f:
        f.write(np.concatenate(chunks).tobytes())
    os.replace(tmp, blob_path)
    with open(man_path + '.tmp', 'w') as f:
        json.dump({'format': 'scs-params-v1', 'dtype': '<f4', 'total': off, 'layers': manifest}, f, indent=0)
    os.replace(man_path + '.tmp', man_path)
    if path.endswith('.pkl'):
        write_reference_pickle(path, params)          # what the reference's load_params_from reads (nets.py:251)
    return blob_path


def load_params(path):
    """Load parameters from our blob+manifest pair, or from a reference ``.pkl``.

    Returns OrderedDict{name: [float32 arrays]} validated against LAYER_SPEC.
    Raises IOError if nothing loadable exists, ValueError on a schema mismatch.
    """
    blob_path, man_path = _paths(path)
    if os.path.exists(blob_path) and os.path.exists(man_path):
        with open(man_path) as f:
            man = json.load(f)
        if man.get('format') != 'scs-params-v1':
            raise ValueError('unknown manifest format in %s' % man_path)
        blob = np.fromfile(blob_path, dtype='<f4')
        if blob.size != man['total']:
            raise ValueError('%s holds %d floats, manifest says %d' % (blob_path, blob.size, man['total']))
        params = OrderedDict()
        for entry in man['layers']:
            arrs = []
            for a in entry['arrays']:
                n = int(np.prod(a['shape']))
                arrs.append(blob[a['offset']:a['offset'] + n].reshape(a['shape']).astype(np.float32))
            params[entry['name']] = arrs
        return check_params(params)
    if path.endswith('.pkl') and os.path.exists(path):
        raw = read_reference_pickle(path)
        params = OrderedDict((k, [np.asarray(a, dtype=np.float32) for a in v]) for k, v in raw.items())
        return check_params(params)
    raise IOError('no parameter file at %s (looked for %s + %s, or a reference .pkl)' % (path, blob_path, man_path))


# ---------------------------------------------------------------------------------------------
# initialisation (Lasagne defaults; used when training from scratch)
# ---------------------------------------------------------------------------------------------
def init_params(seed=0):
    """Lasagne default initialisers: GlorotUniform W (conv & dense), b = 0, BN beta 0 / gamma 1 /
    mean 0 / inv_std 1, PReLU alpha 0.25 (lasagne.layers.{conv,dense,normalization,special})."""
    rng = np.random.RandomState(seed)
    params = OrderedDict()
    for name, shapes in LAYER_SPEC:
        arrs = []
        if name.endswith('_bn'):
            c = shapes[0]
            arrs = [np.zeros(c, np.float32), np.ones(c, np.float32), np.zeros(c, np.float32), np.ones(c, np.float32)]
        elif 'prelu' in name:
            arrs = [np.full(shapes[0], 0.25, np.float32)]
        elif len(shapes) == 1:                       # conv W
            cout, cin, kh, kw = shapes[0]
            lim = np.sqrt(6.0 / ((cin + cout) * kh * kw))    # Glorot: receptive field * (fan_in + fan_out)
            arrs = [rng.uniform(-lim, lim, shapes[0]).astype(np.float32)]
        elif len(shapes) == 2:                       # dense W, b
            nin, nout = shapes[0]
            lim = np.sqrt(6.0 / (nin + nout))
            arrs = [rng.uniform(-lim, lim, shapes[0]).astype(np.float32), np.zeros(shapes[1], np.float32)]
        params[name] = arrs
    return params


def params_to_flat(params):
    """Concatenate all arrays in LAYER_SPEC order -> float32 [N_STORED] (the Lasagne-layout arena)."""
    return np.concatenate([np.asarray(a, np.float32).ravel() for n, _ in LAYER_SPEC for a in params[n]])


def flat_to_params(flat):
    flat = np.asarray(flat, np.float32)
    if flat.size != N_STORED:
        raise ValueError('flat parameter vector has %d floats, expected %d' % (flat.size, N_STORED))
    params, off = OrderedDict(), 0
    for name, shapes in LAYER_SPEC:
        arrs = []
        for s in shapes:
            n = int(np.prod(s))
            arrs.append(flat[off:off + n].reshape(s).copy())
            off += n
        params[name] = arrs
    return params


def head_first_float():
    """First float of the dense head (FC1 .. out_layer: the contiguous tail of the arena) - the second bucket of the
    two-bucket gradient exchange (csrc/scs_common.h head_off().fc1W; scs_wait_head_grads returns the same number)."""
    return flat_offsets()[('FC1', 0)][0]


def flat_offsets():
    """{(layer_name, array_index): (offset, shape)} into the flat arena."""
    out, off = {}, 0
    for name, shapes in LAYER_SPEC:
        for i, s in enumerate(shapes):
            out[(name, i)] = (off, tuple(s))
            off += int(np.prod(s))
    return out
