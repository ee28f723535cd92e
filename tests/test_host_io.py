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
    with pytest.raises(IOError):                                          # whole-stream gather: the member's CRC-32 is verified
        _hip.gz_gather(str(tmp_path / 'flip.gz'), last, 4, total=True)


def test_crc32_of_every_member_is_verified_on_whole_reads(tmp_path, monkeypatch):
    """A payload byte changed in a way that keeps the deflate stream decodable and its length right (a stored block: the bytes
    are the data) is caught by the CRC-32 of the member - in scs_gz_read (what nifti.load uses for T1 / mask volumes) and in a
    gather over the whole stream; SCS_GZ_NOCRC=1 skips the check; a gather that stops before the end of the stream cannot see
    the trailer and returns the (here: untouched) leading bytes."""
    import zlib
    rng = np.random.RandomState(5)
    data = (rng.rand(70000) < 0.02).astype(np.uint8).tobytes() + rng.bytes(3000) + bytes(200000)      # zero runs + noise + a long run
    for level in (0, 6):
        co = zlib.compressobj(level, zlib.DEFLATED, 31)
        blob = bytearray(co.compress(data) + co.flush())
        good = str(tmp_path / ('ok%d.gz' % level))
        open(good, 'wb').write(bytes(blob))
        out = np.empty(len(data), np.uint8)
        _hip.gz_read(good, 0, out)
        assert out.tobytes() == data
        if level == 0:
            # stored blocks: byte 10 + 5 + k of the file is byte k of the data
            blob[10 + 5 + 71000] ^= 0x01
        else:
            blob[-8] ^= 0x01                                            # the trailer's CRC itself
        bad = str(tmp_path / ('bad%d.gz' % level))
        open(bad, 'wb').write(bytes(blob))
        with pytest.raises(IOError, match='CRC'):
            _hip.gz_read(bad, 0, np.empty(len(data), np.uint8))
        with pytest.raises(IOError, match='CRC'):
            _hip.gz_gather(bad, np.array([0], np.int64), 1, total=True)
        head = _hip.gz_gather(bad, np.array([0, 100], np.int64), 4)      # stops after offset 104: no trailer, no check
        assert head.tobytes() == data[0:4] + data[100:104]
        monkeypatch.setenv('SCS_GZ_NOCRC', '1')
        _hip.gz_read(bad, 0, out)                                       # unchecked: the (possibly wrong) bytes come back
        monkeypatch.delenv('SCS_GZ_NOCRC')


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


def test_sparse_gzip_member_round_trips_through_zlib():
    """scs_gz_member (the run / 8-byte-repeat deflate writer of libscs_io.so) against zlib's and gzip's inflaters: every zero-run
    length around the 258-byte match limit, runs next to data, constants, float64 values that came from float32, dense random
    bytes (no compression, still valid), and 300 random structured buffers."""
    import gzip
    import zlib
    from cnn_cort import _hip
    rng = np.random.RandomState(5)

    def check(b):
        b = bytes(b)
        m = _hip.gz_member(b)
        assert zlib.decompress(m, 31) == b and gzip.decompress(m) == b
        return len(m)

    for n in list(range(0, 20)) + list(range(250, 275)) + list(range(510, 530)) + [1000, 4096, 65535, 65536, 100003]:
        check(b'\x00' * n)
        check(b'\x07' + b'\x00' * n + b'\x09')
        check(b'\x05' * n)
    assert check(b'\x00' * (1 << 22)) < 30000                              # 13 bits per 258 bytes
    a = np.zeros(200000, np.float64)
    a[60000:90000] = rng.dirichlet(np.ones(15) * 0.3, 30000)[:, 0].astype(np.float32)
    assert check(a.tobytes()) < 0.8 * 30000 * 8                            # the three zero mantissa bytes of every value are matches
    assert check(np.full(100001, 3, np.uint8).tobytes()) < 1000
    assert check(np.full(70000, -7, np.int16).tobytes()) < 1500
    dense = rng.bytes(100000)
    assert check(dense) < 100000 * 9 // 8 + 64                             # literals only: 8 - 9 bits per byte
    for k in range(300):
        n = rng.randint(0, 3000)
        b = bytearray(n)
        for _ in range(rng.randint(0, 6)):
            if n < 2:
                break
            lo = rng.randint(0, n - 1)
            hi = min(n, lo + rng.randint(1, 400))
            b[lo:hi] = rng.bytes(hi - lo) if rng.rand() < 0.5 else bytes([rng.randint(0, 256)]) * (hi - lo)
        check(b)
    # concatenated members are one stream for the library's own inflate too
    with pytest.raises(Exception):
        _hip.gz_member(np.zeros((4, 4))[:, ::2])                           # not C-contiguous


def test_sparse_volume_files_are_the_same_volume_with_either_encoder(tmp_path, monkeypatch):
    """nifti.save_sparse through the library's encoder and through zlib (SCS_GZIP=zlib): different gzip bytes, the same volume -
    read back with the native inflate and with the Python streaming reader."""
    from cnn_cort import nifti
    rng = np.random.RandomState(6)
    shape = (40, 36, 30, 15)
    vox = np.argwhere(rng.rand(*shape[:3]) > 0.9)
    vals = rng.dirichlet(np.ones(15) * 0.3, len(vox)).astype(np.float32).astype(np.float64)
    nifti.save_sparse(str(tmp_path / 'native.nii.gz'), shape, np.float64, vox, vals)
    monkeypatch.setenv('SCS_GZIP', 'zlib')
    nifti.save_sparse(str(tmp_path / 'zlib.nii.gz'), shape, np.float64, vox, vals)
    monkeypatch.delenv('SCS_GZIP')
    a = nifti.load(str(tmp_path / 'native.nii.gz')).get_data()
    b = nifti.load(str(tmp_path / 'zlib.nii.gz')).get_data()
    monkeypatch.setenv('SCS_NO_NATIVE_IO', '1')
    c = nifti.load(str(tmp_path / 'native.nii.gz')).get_data()
    dense = np.zeros(shape, np.float64)
    dense[vox[:, 0], vox[:, 1], vox[:, 2]] = vals
    assert np.array_equal(a, dense) and np.array_equal(b, dense) and np.array_equal(c, dense)
    assert open(str(tmp_path / 'native.nii.gz'), 'rb').read() != open(str(tmp_path / 'zlib.nii.gz'), 'rb').read()


def test_sparse_member_from_index_and_values():
    """scs_gz_member_scatter: the member of `n` fill elements with values at given indices equals the member's worth of a NumPy
    array built the same way - for a zero and a non-zero fill, several element sizes, duplicate indices (the later one wins) -
    and an index outside the member is an error, not a write."""
    import zlib
    rng = np.random.RandomState(7)
    for dtype, fill in ((np.float64, 0.0), (np.int16, 0), (np.int16, 9), (np.uint8, 3), (np.float32, -1.5)):
        n = 50000
        idx = np.sort(rng.choice(n, 700, replace=False)).astype(np.int64)
        idx = np.concatenate([idx, idx[:5]])                                 # duplicates
        vals = (rng.rand(idx.size) * 100).astype(dtype)
        ref = np.full(n, fill, dtype=dtype)
        ref[idx] = vals
        m = _hip.gz_member_scatter(n, dtype(fill), idx, vals)
        assert zlib.decompress(m, 31) == ref.tobytes(), dtype
    assert zlib.decompress(_hip.gz_member_scatter(10, np.float64(0), np.zeros(0, np.int64), np.zeros(0, np.float64)), 31) == bytes(80)
    with pytest.raises(IOError):
        _hip.gz_member_scatter(100, np.int16(0), np.array([100], np.int64), np.array([1], np.int16))
    with pytest.raises(ValueError):
        _hip.gz_member_scatter(100, np.int16(0), np.array([1, 2], np.int64), np.array([1], np.int16))
