"""
Minimal NIfTI-1 single-file reader / writer (.nii, .nii.gz).

The reference reads and writes volumes through nibabel (/root/reference/cnn_cort/base.py:4-5,413,446-455),
which is not available here and must not be assumed on a GPU box; the hot path only needs: load the voxel
array in (x,y,z[,t]) index order with scl_slope/inter applied the way ``get_data()`` does, keep the header's
geometry, and write an array back with that geometry.
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64,
           256: np.int8, 512: np.uint16, 768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


class NiftiImage:
    def __init__(self, data, header_bytes=None, endian='<'):
        self._data = data
        self.header_bytes = header_bytes
        self.endian = endian

    @property
    def shape(self):
        return self._data.shape

    def get_data(self):
        return self._data

    @property
    def affine(self):
        """4x4 voxel->world matrix from the sform (or pixdim scaling when no sform is set)."""
        if self.header_bytes is None:
            return np.eye(4)
        e = self.endian
        sform_code = struct.unpack(e + 'h', self.header_bytes[254:256])[0]
        if sform_code > 0:
            rows = np.frombuffer(self.header_bytes[280:328], dtype=e + 'f4').reshape(3, 4)
            return np.vstack([rows, [0, 0, 0, 1]]).astype(np.float64)
        pixdim = np.frombuffer(self.header_bytes[76:108], dtype=e + 'f4')
        return np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0]).astype(np.float64)


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith('.gz') else open(path, mode)


def load(path):
    with _open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 352:
        raise IOError('%s: too short for a NIfTI-1 file' % path)
    endian = '<'
    if struct.unpack('<i', raw[0:4])[0] != 348:
        if struct.unpack('>i', raw[0:4])[0] != 348:
            raise IOError('%s: not a NIfTI-1 file (sizeof_hdr != 348)' % path)
        endian = '>'
    if raw[344:348] not in (b'n+1\x00',):
        raise IOError('%s: only single-file NIfTI-1 (magic n+1) is supported' % path)
    dim = struct.unpack(endian + '8h', raw[40:56])
    ndim = dim[0]
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    datatype = struct.unpack(endian + 'h', raw[70:72])[0]
    if datatype not in _DTYPES:
        raise IOError('%s: unsupported NIfTI datatype code %d' % (path, datatype))
    vox_offset = int(struct.unpack(endian + 'f', raw[108:112])[0])
    slope, inter = struct.unpack(endian + '2f', raw[112:120])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=n, offset=vox_offset).reshape(shape, order='F')
    data = np.ascontiguousarray(data.astype(dt.newbyteorder('=')))
    if np.isfinite(slope) and slope != 0 and not (slope == 1.0 and inter == 0.0):
        data = data.astype(np.float64) * slope + inter        # nibabel get_data(): scaled -> float64
    return NiftiImage(data, header_bytes=raw[:348], endian=endian)


def save(data, path, like=None, affine=None):
    """Write `data` (<= 5-D) as NIfTI-1.  Geometry is copied from `like` (a NiftiImage) when given, else built
    from `affine` (default identity)."""
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    code = _CODES.get(data.dtype.str[1:])
    if code is None:
        raise ValueError('cannot store dtype %s in NIfTI-1' % data.dtype)
    hdr = bytearray(348)
    if like is not None and like.header_bytes is not None and like.endian == '<':
        hdr[:] = like.header_bytes
    else:
        A = np.eye(4) if affine is None else np.asarray(affine, np.float64)
        if like is not None:
            A = like.affine
        struct.pack_into('<h', hdr, 254, 2)                                   # sform_code = aligned
        hdr[280:328] = A[:3].astype('<f4').tobytes()
        vox = np.sqrt((A[:3, :3] ** 2).sum(0))
        struct.pack_into('<8f', hdr, 76, 1.0, vox[0], vox[1], vox[2], 1.0, 1.0, 1.0, 1.0)
        hdr[123] = 10                                                         # xyzt_units: mm + s
    struct.pack_into('<i', hdr, 0, 348)
    dims = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into('<8h', hdr, 40, *dims)
    struct.pack_into('<h', hdr, 70, code)
    struct.pack_into('<h', hdr, 72, data.dtype.itemsize * 8)
    struct.pack_into('<f', hdr, 108, 352.0)
    struct.pack_into('<2f', hdr, 112, 1.0, 0.0)                               # scl_slope, scl_inter
    hdr[344:348] = b'n+1\x00'
    payload = bytes(hdr) + b'\x00\x00\x00\x00' + np.asfortranarray(data).astype(data.dtype.newbyteorder('<')).tobytes(order='F')
    with _open(path, 'wb') as f:
        f.write(payload)
