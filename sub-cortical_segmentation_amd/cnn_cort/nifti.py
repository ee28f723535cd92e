"""
Minimal NIfTI-1 single-file reader / writer (.nii, .nii.gz).

The reference reads and writes volumes through nibabel (/root/reference/cnn_cort/base.py:4-5,413,446-455),
which is not available here and must not be assumed on a GPU box; the hot path only needs: load the voxel
array in (x,y,z[,t]) index order with scl_slope/inter applied the way ``get_data()`` does, keep the header's
geometry, and write an array back with that geometry.

A subject of BASELINE configs[4] moves ~0.8 GB of priors in and ~1.5 GB of float64 probabilities out per scan, so
the file layer is written for throughput: arrays keep the file's Fortran memory order (as nibabel's do; no
transposing copy), .gz input is inflated in one zlib call per member, and .gz output is deflated in parallel as
independent gzip members (a valid RFC-1952 stream that gzip/zcat/nibabel read as one file).
"""
import os
import struct
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

GZIP_LEVEL = 1                     # nibabel's default_compresslevel
_MEMBER_BYTES = 16 << 20           # uncompressed bytes per gzip member

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


def _workers():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                               # container CPU quota (cgroup v2), when one is set
        with open("/sys/fs/cgroup/cpu.max") as _f:
            quota, period = _f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


_STREAM_BYTES = 2 << 20             # inflated bytes handed on per step (cache-sized: the gather reads them once)
_FEED_BYTES = 256 << 10             # compressed bytes fed to zlib per call


def _inflate_stream(f):
    """Yield the inflated content of an open (possibly multi-member) gzip file in pieces of <= _STREAM_BYTES,
    never holding more than one piece (the temporary blocks are recycled by the allocator, so a 1 GB volume costs
    one pass of page touching in its final buffer, not three)."""
    d = zlib.decompressobj(31)
    pending = b''
    while True:
        if not pending:
            pending = f.read(_FEED_BYTES)
            if not pending:
                break
        if d.eof:                                       # next member (zero padding between members is legal)
            pending = pending.lstrip(b'\x00')
            if not pending:
                continue
            d = zlib.decompressobj(31)
        out = d.decompress(pending, _STREAM_BYTES)
        pending = d.unused_data if d.eof else d.unconsumed_tail
        if out:
            yield out
    while not d.eof:                                    # output still buffered inside zlib after the last input
        out = d.decompress(b'', _STREAM_BYTES)
        if not out:
            raise IOError('truncated gzip stream')
        yield out


def _plain_stream(f):
    while True:
        out = f.read(_STREAM_BYTES)
        if not out:
            return
        yield out


def _stream(path, f):
    magic = f.read(2)
    f.seek(0)
    return _inflate_stream(f) if magic == b'\x1f\x8b' else _plain_stream(f)


class _Header:
    def __init__(self, raw, path):
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
        shape = tuple(int(d) for d in dim[1:1 + dim[0]])
        while len(shape) > 3 and shape[-1] == 1:
            shape = shape[:-1]
        datatype = struct.unpack(endian + 'h', raw[70:72])[0]
        if datatype not in _DTYPES:
            raise IOError('%s: unsupported NIfTI datatype code %d' % (path, datatype))
        self.endian, self.shape = endian, shape
        self.vox_offset = int(struct.unpack(endian + 'f', raw[108:112])[0])
        if self.vox_offset < 352:
            raise IOError('%s: vox_offset %d inside the header' % (path, self.vox_offset))
        slope, inter = struct.unpack(endian + '2f', raw[112:120])
        self.scaled = bool(np.isfinite(slope) and slope != 0 and not (slope == 1.0 and inter == 0.0))
        self.slope, self.inter = slope, inter
        self.dtype = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
        self.n = int(np.prod(shape))
        self.raw = bytes(raw[:348])

    def finish(self, data):
        """nibabel get_data(): native byte order; scaled -> float64."""
        if self.endian == '>':
            data = data.astype(self.dtype.newbyteorder('='))
        if self.scaled:
            data = data.astype(np.float64) * self.slope + self.inter
        return data


def _voxel_pieces(path):
    """-> (header, iterator over (first_byte_index, bytes) pieces of the voxel block, in order)."""
    f = open(path, 'rb')
    stream = _stream(path, f)
    head = b''
    for piece in stream:                                # the header always fits the first pieces
        head += piece
        if len(head) >= 352:
            break
    try:
        hdr = _Header(head, path)
    except Exception:
        f.close()
        raise

    def pieces():
        try:
            pos = -hdr.vox_offset                       # byte index relative to the first voxel
            first = head
            for piece in _chain(first, stream):
                lo, hi = pos, pos + len(piece)
                pos = hi
                if hi <= 0:
                    continue
                if lo < 0:
                    piece, lo = piece[-lo:], 0
                yield lo, piece
        finally:
            f.close()
    return hdr, pieces()


def _chain(first, rest):
    yield first
    for p in rest:
        yield p


def load(path):
    native = _native_gather(path)
    if native is not None:                              # the library's own inflate (csrc/host_io.cpp), straight into the final buffer
        from . import _hip
        hdr = native
        buf = _hip.gz_read(path, hdr.vox_offset, np.empty(hdr.n * hdr.dtype.itemsize, dtype=np.uint8))
        data = hdr.finish(buf.view(hdr.dtype)).reshape(hdr.shape, order='F')
        return NiftiImage(data, header_bytes=hdr.raw, endian=hdr.endian)
    hdr, pieces = _voxel_pieces(path)
    nbytes = hdr.n * hdr.dtype.itemsize
    buf = np.empty(nbytes, dtype=np.uint8)              # the only full-size allocation; written once
    got = 0
    for lo, piece in pieces:
        take = min(len(piece), nbytes - lo)
        if take <= 0:
            break
        buf[lo:lo + take] = np.frombuffer(piece, dtype=np.uint8, count=take)
        got = lo + take
    if got < nbytes:
        raise IOError('%s: voxel data truncated' % path)
    # Fortran memory order, as on disk and as nibabel returns it: indexing is [x,y,z(,t)] either way
    data = hdr.finish(buf.view(hdr.dtype)).reshape(hdr.shape, order='F')
    return NiftiImage(data, header_bytes=hdr.raw, endian=hdr.endian)


def _native_gather(path):
    """Header of a gzip NIfTI file when the host-side inflate of libscs_io.so can serve load / load_rows for it (the library is
    built and the file is gzip), else None: the Python streaming path then runs.  SCS_NO_NATIVE_IO=1 forces the Python path."""
    if os.environ.get('SCS_NO_NATIVE_IO'):
        return None
    try:
        from . import _hip
        if not os.path.exists(_hip.IO_LIB_PATH):
            return None
        with open(path, 'rb') as f:
            if f.read(2) != b'\x1f\x8b':
                return None
            f.seek(0)
            head = zlib.decompressobj(31).decompress(f.read(1 << 16), 352)
        _hip.load_io_library()
        return _Header(head, path)
    except (OSError, AttributeError, RuntimeError, zlib.error):
        return None


def load_rows(path, voxels):
    """Values of a 3-D/4-D volume at a list of voxels, WITHOUT materialising the volume: the file is streamed once
    and only the requested entries are kept.  `voxels` int [N,3] (x,y,z); returns [N] (3-D file) or [N,T] (4-D)
    in the dtype get_data() would have.  Equals  load(path).get_data()[x, y, z]  exactly.

    This is how test_scan reads the 15-channel prior (base.py:388-394 needs 15 numbers per ROI voxel out of a
    ~0.8 GB volume)."""
    native = _native_gather(path)
    if native is not None:
        hdr, pieces = native, None
    else:
        hdr, pieces = _voxel_pieces(path)
    v = np.asarray(voxels, dtype=np.int64).reshape(-1, 3)
    shape = hdr.shape
    if len(shape) not in (3, 4):
        raise ValueError('%s: load_rows needs a 3-D or 4-D volume, found shape %r' % (path, shape))
    if v.size and ((v < 0).any() or (v >= np.asarray(shape[:3])).any()):
        raise ValueError('voxel outside the volume')
    nvox, nt = shape[0] * shape[1] * shape[2], (shape[3] if len(shape) == 4 else 1)
    lin = v[:, 0] + shape[0] * (v[:, 1] + shape[1] * v[:, 2])
    order = np.argsort(lin, kind='stable')
    lin_sorted = lin[order]
    # element positions wanted, ascending: channel-major (Fortran order: the last axis is the slowest)
    pos = (np.arange(nt, dtype=np.int64)[:, None] * nvox + lin_sorted[None, :]).ravel()
    item = hdr.dtype.itemsize
    if pieces is None:
        # the library's own inflate (csrc/host_io.cpp): the stream passes through a cache-sized window, long zero runs are
        # memsets, only the requested elements are copied out
        from . import _hip
        out = _hip.gz_gather(path, hdr.vox_offset + pos * item, item).view(hdr.dtype).reshape(-1)
        rows = np.empty((v.shape[0], nt), dtype=out.dtype)
        rows[order] = out.reshape(nt, -1).T
        rows = hdr.finish(rows)
        return rows if len(shape) == 4 else rows[:, 0]
    out = np.empty(pos.size, dtype=hdr.dtype)
    done, carry, carry_lo = 0, b'', 0
    for lo, piece in pieces:
        if done >= pos.size:
            break
        if carry:                                       # an element straddling two pieces
            piece, lo = carry + piece, carry_lo
        e0 = -(-lo // item)                             # first whole element in this piece
        skip = e0 * item - lo
        ne = (len(piece) - skip) // item
        tail = skip + ne * item
        carry, carry_lo = piece[tail:], lo + tail
        if ne <= 0:
            continue
        hi = np.searchsorted(pos, e0 + ne, side='left')
        if hi > done:
            arr = np.frombuffer(piece, dtype=hdr.dtype, count=ne, offset=skip)
            out[done:hi] = arr[pos[done:hi] - e0]
            done = hi
    if done < pos.size:
        raise IOError('%s: voxel data truncated' % path)
    rows = np.empty((v.shape[0], nt), dtype=out.dtype)
    rows[order] = out.reshape(nt, -1).T
    rows = hdr.finish(rows)
    return rows if len(shape) == 4 else rows[:, 0]


def _header_bytes(data_shape, dtype, like, affine):
    code = _CODES.get(np.dtype(dtype).str[1:])
    if code is None:
        raise ValueError('cannot store dtype %s in NIfTI-1' % dtype)
    if len(data_shape) > 7:
        raise ValueError('NIfTI-1 stores at most 7 dimensions')
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
    dims = [len(data_shape)] + list(data_shape) + [1] * (7 - len(data_shape))
    struct.pack_into('<8h', hdr, 40, *dims)
    struct.pack_into('<h', hdr, 70, code)
    struct.pack_into('<h', hdr, 72, np.dtype(dtype).itemsize * 8)
    struct.pack_into('<f', hdr, 108, 352.0)
    struct.pack_into('<2f', hdr, 112, 1.0, 0.0)                               # scl_slope, scl_inter
    hdr[344:348] = b'n+1\x00'
    return bytes(hdr) + b'\x00\x00\x00\x00'


def _gzip_member(chunk, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(GZIP_LEVEL, zlib.DEFLATED, 31, 8, strategy)
    return c.compress(chunk) + c.flush()


def _write_gzip(path, buffers):
    """Write the concatenation of `buffers` (bytes-like, C-contiguous) as parallel-deflated gzip members."""
    chunks = []
    for b in buffers:
        mv = memoryview(b).cast('B')
        chunks.extend(mv[o:o + _MEMBER_BYTES] for o in range(0, len(mv), _MEMBER_BYTES))
    with open(path, 'wb') as f, ThreadPoolExecutor(_workers()) as pool:
        for member in pool.map(_gzip_member, chunks):   # zlib releases the GIL; map keeps the order
            f.write(member)


_SPARSE_STRATEGY = {'default': zlib.Z_DEFAULT_STRATEGY, 'rle': zlib.Z_RLE, 'filtered': zlib.Z_FILTERED}[os.environ.get('SCS_SPARSE_STRATEGY', 'default')]


def _native_sparse():
    """The binding of libscs_io.so when its deflate writer for sparse volumes serves (csrc/host_io.cpp: scs_gz_member[_scatter] -
    zero runs and 8-byte repeats as matches, the rest literals; any inflater reads it; a fraction of zlib level 1's CPU time on
    rows that are mostly zeros, ~25 % larger on the data they hold), None when the library is not built or SCS_GZIP=zlib."""
    if os.environ.get('SCS_GZIP', '') == 'zlib' or os.environ.get('SCS_NO_NATIVE_IO'):
        return None
    from . import _hip
    return _hip if os.path.exists(_hip.IO_LIB_PATH) else None


_GAP_ROWS = 64                     # data rows closer than this share a gzip member (the zeros between them are deflated with them)
_MAX_ROWS_BYTES = 16 << 20         # largest member


def save_sparse(path, shape, dtype, voxels, values, like=None, affine=None, fill=0):
    """Write a volume that is `fill` (default zero) except at `voxels` (int [N,3]) where it holds `values` ([N] for a 3-D shape,
    [N,T] for a 4-D one), without building the dense array.  The file is, after gunzip, byte for byte what save() writes for
    the equivalent dense array.  Later duplicates of a voxel win, as in a fancy assignment.

    The voxel block is cut into gzip members along its x-rows: runs of rows that hold data (rows closer than _GAP_ROWS joined)
    are scattered into a scratch buffer and deflated - in parallel - and every run of all-zero rows in between is written as a
    few cached members of 2^k zero rows, deflated once per size and process.  Only the rows that hold data go through the
    compressor: for the float64 [256,256,192,15] probability volume of test_scan (1.5 GB dense, ~2.6 % of its elements set)
    that is 0.3 GB instead of the 1 GB that fixed 16 MB members cost (1.6 -> 0.5 CPU-seconds per subject).

    test_scan uses it for the float64 [X,Y,Z,15] probability volume."""
    dtype = np.dtype(dtype)
    shape = tuple(int(d) for d in shape)
    if len(shape) not in (3, 4):
        raise ValueError('save_sparse writes 3-D or 4-D volumes')
    v = np.asarray(voxels, dtype=np.int64).reshape(-1, 3)
    if v.size and ((v < 0).any() or (v >= np.asarray(shape[:3])).any()):
        raise ValueError('voxel outside the volume')
    nvox, nt = shape[0] * shape[1] * shape[2], (shape[3] if len(shape) == 4 else 1)
    vals = np.asarray(values).reshape(v.shape[0], nt)
    lin = v[:, 0] + shape[0] * (v[:, 1] + shape[1] * v[:, 2])
    order = np.argsort(lin, kind='stable')
    pos = (np.arange(nt, dtype=np.int64)[:, None] * nvox + lin[order][None, :]).ravel()
    flat_vals = np.ascontiguousarray(vals[order].T, dtype=dtype.newbyteorder('<')).ravel()
    head = _header_bytes(shape, dtype, like, affine)
    row = shape[0]                                      # elements per x-row (the fastest axis)
    total_rows = nvox * nt // row
    max_rows = max(1, _MAX_ROWS_BYTES // (row * dtype.itemsize))
    # segments in file order: ('z', first row, rows) all-zero / ('d', first row, rows, first value, end value) holding data
    segments = []
    if pos.size:
        prow = pos // row                               # non-decreasing: pos ascends
        data_rows = prow[np.concatenate([[True], prow[1:] != prow[:-1]])]
        cut = np.nonzero(np.diff(data_rows) > _GAP_ROWS)[0]
        run_lo = data_rows[np.concatenate([[0], cut + 1])]
        run_hi = data_rows[np.concatenate([cut, [data_rows.size - 1]])] + 1
        at = 0
        for lo, hi in zip(run_lo.tolist(), run_hi.tolist()):
            if lo > at:
                segments.append(('z', at, lo - at))
            for r0 in range(lo, hi, max_rows):
                r1 = min(hi, r0 + max_rows)
                a, b = np.searchsorted(prow, [r0, r1])
                segments.append(('d', r0, r1 - r0, int(a), int(b)))
            at = hi
        if at < total_rows:
            segments.append(('z', at, total_rows - at))
    else:
        segments.append(('z', 0, total_rows))

    if not str(path).endswith('.gz'):
        with open(path, 'wb') as f:
            f.write(head)
            for seg in segments:
                if seg[0] == 'z':
                    left = seg[2] * row * dtype.itemsize
                    zeros = np.full(min(left, 4 << 20) // dtype.itemsize, fill, dtype=dtype.newbyteorder('<')).tobytes()
                    while left > 0:
                        f.write(zeros[:min(left, len(zeros))])
                        left -= min(left, len(zeros))
                else:
                    _, r0, nr, a, b = seg
                    buf = np.full(nr * row, fill, dtype=dtype.newbyteorder('<'))
                    buf[pos[a:b] - r0 * row] = flat_vals[a:b]
                    f.write(memoryview(buf).cast('B'))
        return

    native = _native_sparse()

    def data_member(seg):
        _, r0, nr, a, b = seg
        if native is not None:                          # buffer, scatter and deflate inside libscs_io.so
            return native.gz_member_scatter(nr * row, np.asarray(fill, dtype=flat_vals.dtype), pos[a:b] - r0 * row, flat_vals[a:b])
        buf = np.full(nr * row, fill, dtype=dtype.newbyteorder('<'))
        buf[pos[a:b] - r0 * row] = flat_vals[a:b]
        return _gzip_member(memoryview(buf).cast('B'), _SPARSE_STRATEGY)

    fill_bytes = np.asarray(fill, dtype=dtype.newbyteorder('<')).tobytes()

    def zero_members(nrows):
        """nrows rows of the fill value as cached members of 2^k rows each (largest first)."""
        out = []
        k = max_rows.bit_length() - 1
        while nrows > 0:
            while (1 << k) > nrows:
                k -= 1
            out.append(_zero_member((1 << k) * row * dtype.itemsize, fill_bytes))
            nrows -= 1 << k
        return out

    # a few hundred data segments per file: handed to the pool in batches (one task per ~2 MB of rows), written in few large writes
    dsegs = [sg for sg in segments if sg[0] == 'd']
    batches, cur, cur_bytes = [], [], 0
    for sg in dsegs:
        cur.append(sg)
        cur_bytes += sg[2] * row * dtype.itemsize
        if cur_bytes >= (2 << 20):
            batches.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        batches.append(cur)
    with open(path, 'wb') as f, ThreadPoolExecutor(_workers()) as pool:
        done = pool.map(lambda b: [data_member(sg) for sg in b], batches)          # zlib releases the GIL; map keeps the order
        ready = iter(())
        out, out_bytes = [_gzip_member(head)], 0
        for seg in segments:
            if seg[0] == 'z':
                ms = zero_members(seg[2])
            else:
                m = next(ready, None)
                if m is None:
                    ready = iter(next(done))
                    m = next(ready)
                ms = [m]
            out.extend(ms)
            out_bytes += sum(len(m) for m in ms)
            if out_bytes >= (4 << 20):
                f.write(b''.join(out))
                out, out_bytes = [], 0
        f.write(b''.join(out))


_zero_cache = {}
_zero_lock = threading.Lock()


def _zero_member(nbytes, pattern=b'\x00'):
    """A gzip member that inflates to `nbytes` bytes of `pattern` repeated (zeros by default; deflated once per size, pattern
    and process)."""
    key = (nbytes, pattern if any(pattern) else b'\x00')
    m = _zero_cache.get(key)
    if m is None:
        with _zero_lock:
            m = _zero_cache.get(key)
            if m is None:
                m = _zero_cache[key] = _gzip_member(bytes(nbytes) if not any(pattern) else pattern * (nbytes // len(pattern)))
    return m


def save_mostly_constant(data, path, like=None, affine=None):
    """save(data, path) for a volume that is one value almost everywhere (a label map: background, a few structures) - through
    save_sparse with that value as the fill, so that only the rows which differ are deflated (0.09 -> 0.02 CPU-seconds for a
    256x256x192 int16 segmentation).  Falls back to save() when more than 5 % of the voxels differ from the first one."""
    data = np.asarray(data)
    if data.ndim != 3 or data.dtype == np.bool_ or not str(path).endswith('.gz'):
        return save(data, path, like, affine)
    fill = data.flat[0]
    diff = np.nonzero(data != fill)
    if diff[0].size > 0.05 * data.size:
        return save(data, path, like, affine)
    vox = np.stack(diff, axis=1)
    return save_sparse(path, data.shape, data.dtype, vox, data[diff], like=like, affine=affine, fill=fill)


def save(data, path, like=None, affine=None):
    """Write `data` (<= 7-D) as NIfTI-1.  Geometry is copied from `like` (a NiftiImage) when given, else built
    from `affine` (default identity)."""
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    head = _header_bytes(data.shape, data.dtype, like, affine)
    # voxel bytes in Fortran order = the C-order bytes of the transposed array (no copy when data is F-contiguous)
    vox = np.ascontiguousarray(data.T, dtype=data.dtype.newbyteorder('<'))
    if str(path).endswith('.gz'):
        _write_gzip(path, [head, vox])
    else:
        with open(path, 'wb') as f:
            f.write(head)
            f.write(memoryview(vox).cast('B'))
