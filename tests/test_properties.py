"""Property tests (hypothesis) of the host-side restatements against the literal procedures in oracle/patches_ref.py:
arbitrary volume shapes, memory orders, dtypes and centre lists incl. every border case."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from cnn_cort import base, nifti
from oracle import patches_ref as R

shapes = st.tuples(st.integers(1, 9), st.integers(1, 9), st.integers(1, 9))


@settings(max_examples=60, deadline=None)
@given(shape=shapes, seed=st.integers(0, 2 ** 31 - 1), order=st.sampled_from(['C', 'F']),
       mode=st.sampled_from(['axial', 'coronal', 'saggital']), dtype=st.sampled_from([np.float32, np.float64, np.int16]),
       psize=st.sampled_from([(32, 32), (8, 8), (6, 4)]))
def test_get_patches_equals_the_literal_slicing(shape, seed, order, mode, dtype, psize):
    rng = np.random.RandomState(seed)
    vol = np.asarray((rng.rand(*shape) * 100).astype(dtype), order=order)
    n = rng.randint(0, 12)
    centers = np.stack([rng.randint(0, s, n) for s in shape], axis=1) if n else np.zeros((0, 3), np.int64)
    got = base.get_patches(vol, centers, psize, mode=mode)
    want = R.get_patches(vol, [tuple(c) for c in centers], psize, mode=mode)
    assert got.shape == (n,) + tuple(psize) and got.dtype == vol.dtype
    for g, w in zip(got, want):
        assert np.array_equal(g, w.reshape(psize))           # np.squeeze drops unit axes of tiny patches too


@settings(max_examples=40, deadline=None)
@given(shape=st.tuples(st.integers(2, 8), st.integers(2, 8), st.integers(2, 8)), seed=st.integers(0, 2 ** 31 - 1),
       nclass=st.integers(1, 14), density=st.floats(0.0, 1.0))
def test_post_processing_equals_the_literal_procedure(shape, seed, nclass, density):
    rng = np.random.RandomState(seed)
    seg = (rng.randint(1, nclass + 1, shape) * (rng.rand(*shape) < density)).astype(np.int16)
    atlas = rng.rand(*shape) < 0.4
    assert np.array_equal(base.post_process_mask(seg, atlas), R.post_process_segmentation(seg, atlas))


@settings(max_examples=25, deadline=None)
@given(shape=st.tuples(st.integers(1, 7), st.integers(1, 7), st.integers(1, 7), st.integers(2, 4)), seed=st.integers(0, 2 ** 31 - 1),
       gz=st.booleans(), dtype=st.sampled_from([np.float32, np.float64, np.int16, np.uint8]))
def test_nifti_rows_and_sparse_agree_with_dense(tmp_path_factory, shape, seed, gz, dtype):
    # (a trailing 4th dimension of 1 is dropped on load - a T1 stored as [X,Y,Z,1] reads as 3-D - hence >= 2 here)
    rng = np.random.RandomState(seed)
    d = tmp_path_factory.mktemp('nii')
    ext = '.nii.gz' if gz else '.nii'
    a = (rng.rand(*shape) * 50).astype(dtype)
    nifti.save(a, str(d / ('a' + ext)))
    n = rng.randint(0, 10)
    vox = np.stack([rng.randint(0, s, n) for s in shape[:3]], axis=1) if n else np.zeros((0, 3), np.int64)
    rows = nifti.load_rows(str(d / ('a' + ext)), vox)
    assert np.array_equal(rows, a[vox[:, 0], vox[:, 1], vox[:, 2]])
    assert np.array_equal(nifti.load(str(d / ('a' + ext))).get_data(), a)
    vals = (rng.rand(n, shape[3]) * 9 + 1).astype(dtype)
    dense = np.zeros(shape, dtype)
    dense[vox[:, 0], vox[:, 1], vox[:, 2]] = vals
    nifti.save_sparse(str(d / ('s' + ext)), shape, dtype, vox, vals)
    assert np.array_equal(nifti.load(str(d / ('s' + ext))).get_data(), dense)
