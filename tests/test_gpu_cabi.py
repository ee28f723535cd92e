"""The boundary is a C ABI: plain-C programs (gcc, include/scs.h, -lscs; no Python or torch on their side) drive
scs_forward_host (host buffers, no HIP headers either) and scs_train_step / scs_adam_apply (device buffers from the HIP
runtime's C API) and must produce the same bits as the Python binding on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


def lcg_units(count, state=12345):
    """The generator of tests/c/abi_host.c, vectorised: x_{k+1} = a x_k + c (mod 2^32); value = (x >> 8) / 2^24."""
    a, c, out = 1664525, 1013904223, np.empty(count, np.uint32)
    # jump-ahead in blocks: x_{k+j} = A_j x_k + C_j
    block = 1 << 16
    A, C = np.empty(block, np.uint64), np.empty(block, np.uint64)
    Ak, Ck = 1, 0
    for j in range(block):
        Ak, Ck = (Ak * a) & 0xFFFFFFFF, (Ck * a + c) & 0xFFFFFFFF
        A[j], C[j] = Ak, Ck
    x = state
    for lo in range(0, count, block):
        m = min(block, count - lo)
        vals = (A[:m] * np.uint64(x) + C[:m]) & np.uint64(0xFFFFFFFF)
        out[lo:lo + m] = vals.astype(np.uint32)
        x = int(vals[m - 1])
    return (out >> 8).astype(np.float32) / np.float32(16777216.0), x


def test_plain_c_host_matches_the_python_binding(tmp_path, handle):
    exe = str(tmp_path / 'abi_host')
    lib_dir = os.path.join(PKG, 'lib')
    subprocess.check_call(['gcc', '-O1', '-std=c99', '-Wall', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'c', 'abi_host.c'), '-o', exe,
                           '-L', lib_dir, '-lscs', '-Wl,-rpath,' + lib_dir])
    n = 300
    out = str(tmp_path / 'out.bin')
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ':/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    res = subprocess.run([exe, os.path.join(PKG, 'nets', 'miccai2012_v1', 'miccai2012_v1.f32'), str(n), out],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    probs_c = raw[:n * 60].view(np.float32).reshape(n, 15)
    labels_c = raw[n * 60:].view(np.int32)
    # the same inputs in NumPy
    state, views = 12345, []
    for _ in range(3):
        u, state = lcg_units(1024 * n, state)
        views.append((np.float32(4.0) * u - np.float32(2.0)).reshape(n, 1, 32, 32))
    u, state = lcg_units(15 * n, state)
    atlas = u.reshape(n, 15)
    s = np.zeros(n, np.float32)
    for c in range(15):                                      # the C loop's summation order
        s = s + atlas[:, c]
    atlas = atlas / s[:, None]
    probs = np.empty((n, 15), np.float32)
    labels = np.empty((n,), np.int32)
    handle.forward_host(views[0], views[1], views[2], atlas, n, probs, labels)
    assert np.array_equal(probs, probs_c) and np.array_equal(labels, labels_c)
    assert np.abs(probs_c.sum(1) - 1).max() < 1e-5


def test_plain_c_training_host_matches_the_python_binding(tmp_path):
    """A C host (gcc -std=c99, include/scs.h, the HIP runtime's C API for its device buffers) runs three training steps +
    Adam updates; the losses and the whole updated parameter arena equal the Python binding's bit for bit."""
    import torch
    from cnn_cort import _hip
    exe = str(tmp_path / 'abi_train')
    lib_dir = os.path.join(PKG, 'lib')
    subprocess.check_call(['gcc', '-O1', '-std=c99', '-Wall', '-I', os.path.join(ROOT, 'include'), '-I', '/opt/rocm/include',
                           os.path.join(ROOT, 'tests', 'c', 'abi_train.c'), '-o', exe,
                           '-L', lib_dir, '-lscs', '-L', '/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,' + lib_dir])
    n, steps = 96, 3
    out = str(tmp_path / 'train.bin')
    weights_file = os.path.join(PKG, 'nets', 'miccai2012_v1', 'miccai2012_v1.f32')
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ':/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    res = subprocess.run([exe, weights_file, str(n), str(steps), out], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.float32)
    losses_c, params_c = raw[:steps], raw[steps:]
    assert params_c.size == _hip.N_PARAMS
    # the same inputs in NumPy (the C program draws the three views, then per row 15 prior values and the label)
    state, views = 4242, []
    for _ in range(3):
        u, state = lcg_units(1024 * n, state)
        views.append((np.float32(4.0) * u - np.float32(2.0)).reshape(n, 1, 32, 32))
    u, state = lcg_units(16 * n, state)
    u = u.reshape(n, 16)
    atlas = u[:, :15].copy()
    s = np.zeros(n, np.float32)
    for c in range(15):
        s = s + atlas[:, c]
    atlas = atlas / s[:, None]
    y = (u[:, 15] * np.float32(15.0)).astype(np.uint8)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0')
    h = _hip.Handle(0)
    try:
        h.set_params(np.fromfile(weights_file, dtype=np.float32))
        ax, co, sa, at, yd = dev(views[0]), dev(views[1]), dev(views[2]), dev(atlas), dev(y)
        losses = []
        for step in range(steps):
            losses.append(h.train_step(ax, co, sa, at, yd, n, seed=1000 + step))
            h.adam_apply(lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0)
        params = h.get_params()
    finally:
        h.close()
    assert np.array_equal(np.asarray(losses, np.float32), losses_c)
    assert np.array_equal(params, params_c)
    assert np.isfinite(params_c).all() and np.isfinite(losses_c).all() and (losses_c > 0).all()
    assert not np.array_equal(params_c, np.fromfile(weights_file, dtype=np.float32))          # the steps did move the parameters


def test_the_binding_refuses_short_buffers_before_any_launch(handle):
    """The kernels trust the sizes they are given; the ctypes binding checks every buffer against the call's geometry, so a
    short one is a ValueError and not a GPU memory fault."""
    import torch
    dev = 'cuda:0'
    n = 8
    x = torch.zeros((n, 1, 32, 32), device=dev)
    at = torch.zeros((n, 15), device=dev)
    probs = torch.zeros((n, 15), device=dev)
    with pytest.raises(ValueError):
        handle.forward(x, x, x[:4], at, n, probs)
    with pytest.raises(ValueError):
        handle.forward(x, x, x, at, n, probs[:4])
    vol = torch.zeros((16, 16, 16), device=dev)
    centers = torch.zeros((n, 3), dtype=torch.int32, device=dev)
    with pytest.raises(ValueError):
        handle.forward_volume(vol, (16, 16, 32), centers, torch.zeros((16, 16, 16, 15), device=dev), n, probs)
    with pytest.raises(ValueError):
        handle.forward_volume(vol, (16, 16, 16), centers, torch.zeros((16, 16, 8, 15), device=dev), n, probs)
    with pytest.raises(ValueError):
        handle.forward_volume_rows(vol, (16, 16, 16), centers[:4], at, n, probs)
    mask = torch.zeros((16, 16, 16), dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        handle.dilate_mask(mask, mask[:8], (16, 16, 16), 2)
    y = torch.zeros((n,), dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        handle.train_step(x, x, x, at, y[:4], n)
    with pytest.raises(ValueError):
        handle.gather_labels(mask, torch.zeros((1, 3), dtype=torch.int64, device=dev), centers, n, y)     # 3 ints per row: short
