"""Parameter store: layer inventory, file formats, arena layout (CPU)."""
import os
import pickle

import numpy as np
import pytest

from cnn_cort import params as P


def test_layer_inventory_matches_shipped_weights(weights):
    assert len(P.LAYER_SPEC) == 88
    assert sum(1 for _, s in P.LAYER_SPEC if s) == 56
    assert P.N_STORED == 883455
    n_train = sum(int(np.prod(s)) for n, shapes in P.LAYER_SPEC for i, s in enumerate(shapes) if P.is_trainable(n, i))
    assert n_train == 882375
    assert list(weights.keys()) == [n for n, _ in P.LAYER_SPEC]
    # BN inv_std strictly positive, spelling 'saggital' preserved
    assert all((weights['%s_ch_conv%d_bn' % (t, i)][3] > 0).all() for t in P.TOWERS for i in range(1, 6))
    assert 'saggital_d1' in weights


def test_blob_roundtrip(tmp_path, weights):
    p = P.save_params(str(tmp_path / 'w.pkl'), weights)     # '.pkl' stem is accepted, pair is written
    assert p.endswith('w.f32') and os.path.exists(str(tmp_path / 'w.json'))
    back = P.load_params(str(tmp_path / 'w.pkl'))
    for k in weights:
        for a, b in zip(weights[k], back[k]):
            assert a.dtype == np.float32 and np.array_equal(a, b)
    flat = P.params_to_flat(weights)
    again = P.flat_to_params(flat)
    assert all(np.array_equal(a, b) for k in weights for a, b in zip(weights[k], again[k]))
    offs = P.flat_offsets()
    off, shape = offs[('fc_2', 0)]
    assert shape == (555, 270) and np.array_equal(flat[off:off + 555 * 270].reshape(shape), weights['fc_2'][0])


def test_missing_and_mismatched_files_raise(tmp_path, weights):
    with pytest.raises(IOError):
        P.load_params(str(tmp_path / 'nothing.pkl'))
    bad = {k: [a.copy() for a in v] for k, v in weights.items()}
    bad['FC1'][0] = bad['FC1'][0][:, :-1]
    with pytest.raises(ValueError):
        P.check_params(bad)
    with pytest.raises(ValueError):
        P.flat_to_params(np.zeros(10, np.float32))


def test_reference_pickle_loader_is_data_only(tmp_path):
    # a python-2 style pickle of OrderedDict/ndarray loads; anything else is refused
    from collections import OrderedDict
    ok = tmp_path / 'ok.pkl'
    with open(ok, 'wb') as f:
        pickle.dump(OrderedDict(a=[np.arange(3, dtype=np.float32)]), f, protocol=2)
    assert np.array_equal(P.read_reference_pickle(str(ok))['a'][0], [0, 1, 2])
    evil = tmp_path / 'evil.pkl'
    with open(evil, 'wb') as f:
        pickle.dump(os.getcwd, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        P.read_reference_pickle(str(evil))


def test_init_params_follow_lasagne_defaults():
    p = P.init_params(3)
    P.check_params(p)
    assert np.all(p['axial_ch_conv2_bn'][1] == 1) and np.all(p['axial_ch_conv2_bn'][3] == 1)
    assert np.all(p['prelu_f1'][0] == 0.25) and np.all(p['FC1'][1] == 0)
    lim = np.sqrt(6.0 / ((20 + 20) * 9))
    assert np.abs(p['axial_ch_conv2'][0]).max() <= lim


# sha256 / size of the reference's own nets/miccai2012_v1/miccai2012_v1.pkl (a data fixture: the file itself cannot travel)
REF_PKL_SHA256 = '5e3a693f9904b256357bdc863fe4bc2d59630874dd4b356155e75713e30bfcc7'
REF_PKL_BYTES = 3540029


def test_pickle_writer_reproduces_the_reference_weights_file(tmp_path, weights):
    """convert -> save -> the file nolearn's save_params_to wrote (nets.py:155): byte for byte (python-2 cPickle protocol 2,
    numpy 1.x spelling), so the reference's load_params_from (nets.py:251) can read weights written here."""
    import hashlib
    out = str(tmp_path / 'miccai2012_v1.pkl')
    P.write_reference_pickle(out, weights)
    data = open(out, 'rb').read()
    assert len(data) == REF_PKL_BYTES
    assert hashlib.sha256(data).hexdigest() == REF_PKL_SHA256
    ref = '/root/reference/nets/miccai2012_v1/miccai2012_v1.pkl'
    if os.path.exists(ref):                       # authoring container only
        assert data == open(ref, 'rb').read()
    assert b'numpy._core' not in data and b'numpy.core.multiarray' in data     # readable by numpy 1.x
    back = P.read_reference_pickle(out)
    assert list(back.keys()) == [n for n, _ in P.LAYER_SPEC]
    for k in weights:
        assert len(back[k]) == len(weights[k])
        for a, b in zip(weights[k], back[k]):
            assert b.dtype == np.float32 and a.tobytes() == np.ascontiguousarray(b).tobytes()


def test_save_params_to_pkl_writes_a_pickle_the_loader_reads_back(tmp_path):
    p = P.init_params(11)
    p['axial_ch_conv3_bn'][2][:] = np.linspace(-1, 1, 40)             # running statistics travel too
    path = str(tmp_path / 'trained.pkl')
    P.save_params(path, p)
    assert os.path.exists(path) and os.path.exists(str(tmp_path / 'trained.f32'))
    os.remove(str(tmp_path / 'trained.f32'))
    os.remove(str(tmp_path / 'trained.json'))
    back = P.load_params(path)                                        # only the pickle is left
    assert all(np.array_equal(a, b) for k in p for a, b in zip(p[k], back[k]))
    again = str(tmp_path / 'again.pkl')
    P.write_reference_pickle(again, back)
    assert open(again, 'rb').read() == open(path, 'rb').read()       # write -> read -> write is the identity on bytes
    # a path without the .pkl extension writes the pair only
    P.save_params(str(tmp_path / 'plain'), p)
    assert not os.path.exists(str(tmp_path / 'plain.pkl'))
