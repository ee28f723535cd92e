"""The library's host-side inflate (csrc/host_io.cpp, scs_gz_gather): elements of a gzip file at byte offsets of its inflated
content, against Python's zlib on every block type and on the stream shapes NIfTI files come in.  CPU only."""
import gzip
import os
import zlib

import numpy as np
import pytest

from cnn_cort import _hip, nifti


def _write(path, data, level=6, members=()):
    with open(path, 'wb') as f:
        at = 0
        for m in members:
            c = zlib.compressobj(level, zlib.DEFLATED, 31)
            f.write(c.compress(data[at:at + m]) + c.flush() + b'\0' * 3)       # zero padding between members is legal
            at += m
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        f.write(c.compress(data[at:]) + c.flush())


CASES = {
    'zeros': lambda r: bytes(700000),
    'random (literals / stored)': lambda r: r.bytes(400000),
    'sparse float32': lambda r: (np.where(r.rand(300000) < 0.03, r.rand(300000), 0).astype(np.float32)).tobytes(),
    'period 251': lambda r: (np.arange(500000) % 251).astype(np.uint8).tobytes(),
    'period 7': lambda r: (np.arange(500000) % 7).astype(np.uint8).tobytes(),
    'text': lambda r: b'the quick brown fox jumps over the lazy dog. ' * 9000,
    'one byte': lambda r: b'a',
    'empty': lambda r: b'',
}


@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('level,members', [(1, ()), (9, ()), (0, ()), (6, (100003, 17))])
def test_gather_equals_python_zlib(tmp_path, name, level, members):
    rng = np.random.RandomState(len(name) + level)
    data = CASES[name](rng)
    path = str(tmp_path / 'x.gz')
    _write(path, data, level, members if len(data) > 200000 else ())
    with gzip.open(path, 'rb') as f:
        assert f.read() == data                                   # a valid file to begin with
    a = np.frombuffer(data, np.uint8)
    for elem in (1, 2, 4, 8):
        cnt = min(3000, max(0, len(data) - elem + 1))
        off = np.sort(rng.choice(len(data) - elem + 1, cnt, replace=False)).astype(np.int64) if cnt else np.zeros(0, np.int64)
        got, total = _hip.gz_gather(path, off, elem, total=True)
        assert total == len(data)
        assert np.array_equal(got, np.stack([a[o:o + elem] for o in off]) if cnt else np.zeros((0, elem), np.uint8))
        # the contiguous form: a range in the middle, and everything
        lo, hi = len(data) // 3, len(data) - len(data) // 5
        assert _hip.gz_read(path, lo, np.empty(hi - lo, np.uint8)).tobytes() == data[lo:hi]
        assert _hip.gz_read(path, 0, np.empty(len(data), np.uint8)).tobytes() == data
        if len(data):
            with pytest.raises(IOError):
                _hip.gz_read(path, 1, np.empty(len(data), np.uint8))           # one byte beyond the end
        if cnt > 10:                                              # early exit: only the first elements are wanted
            assert np.array_equal(_hip.gz_gather(path, off[:cnt // 3], elem), got[:cnt // 3])


def test_gather_refuses_bad_input(tmp_path):
    data = b'0123456789' * 5000
    path = str(tmp_path / 'x.gz')
    _write(path, data)
    blob = open(path, 'rb').read()
    (tmp_path / 'trunc.gz').write_bytes(blob[:len(blob) // 2])
    (tmp_path / 'notgz.bin').write_bytes(b'\x00' * 100 + data[:100])
    flipped = bytearray(blob)
    flipped[len(blob) // 2] ^= 0x5A
    (tmp_path / 'flip.gz').write_bytes(bytes(flipped))
    last = np.array([len(data) - 4], np.int64)
    with pytest.raises(IOError):
        _hip.gz_gather(str(tmp_path / 'trunc.gz'), last, 4)
    with pytest.raises(IOError):
        _hip.gz_gather(str(tmp_path / 'notgz.bin'), np.array([0], np.int64), 1)
    with pytest.raises(IOError):
        _hip.gz_gather(str(tmp_path / 'missing.gz'), last, 4)
    with pytest.raises(IOError):
        _hip.gz_gather(path, np.array([len(data)], np.int64), 4)          # beyond the end
    with pytest.raises(IOError):
        _hip.gz_gather(path, np.array([8, 4], np.int64), 4)               # not ascending
    try:                                                                  # a flipped bit is either detected or changes the bytes -
        got, total = _hip.gz_gather(str(tmp_path / 'flip.gz'), last, 4, total=True)      # the CRC is deliberately not checked
    except IOError:
        pass


def test_load_rows_native_and_python_paths_agree(tmp_path, monkeypatch):
    """nifti.load_rows through scs_gz_gather and through the Python streaming path: the same rows, for a multi-member file as
    nifti.save writes it and for a single stream as an external tool does."""
    rng = np.random.RandomState(3)
    shape = (23, 19, 17, 15)
    a = np.where(rng.rand(*shape) < 0.1, rng.rand(*shape), 0).astype(np.float32)
    vox = np.stack([rng.randint(0, shape[i], 500) for i in range(3)], 1)
    monkeypatch.setattr(nifti, '_MEMBER_BYTES', 1 << 14)
    nifti.save(a, str(tmp_path / 'multi.nii.gz'))
    with gzip.open(str(tmp_path / 'multi.nii.gz'), 'rb') as f:
        raw = f.read()
    with gzip.open(str(tmp_path / 'single.nii.gz'), 'wb') as f:
        f.write(raw)
    for fn in ('multi.nii.gz', 'single.nii.gz'):
        assert nifti._native_gather(str(tmp_path / fn)) is not None
        native = nifti.load_rows(str(tmp_path / fn), vox)
        whole = nifti.load(str(tmp_path / fn)).get_data()
        monkeypatch.setenv('SCS_NO_NATIVE_IO', '1')
        python = nifti.load_rows(str(tmp_path / fn), vox)
        whole_py = nifti.load(str(tmp_path / fn)).get_data()
        monkeypatch.delenv('SCS_NO_NATIVE_IO')
        assert whole.dtype == whole_py.dtype and whole.flags['F_CONTIGUOUS'] and np.array_equal(whole, whole_py) and np.array_equal(whole, a)
        assert native.dtype == python.dtype == np.float32
        assert np.array_equal(native, python) and np.array_equal(native, a[vox[:, 0], vox[:, 1], vox[:, 2]])
