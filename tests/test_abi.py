"""The C-ABI library loads and exports every symbol include/scs.h declares (no compute without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'scs.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(scs_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from cnn_cort import _hip
    lib = _hip.load_library()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), 'libscs.so does not export %s' % n
    assert set(names) == set(_hip.SIGNATURES), 'ctypes prototypes out of sync with include/scs.h'
    assert lib.scs_abi_version() == 1
    assert lib.scs_target_arch() == b'gfx950'


def test_io_library_exports_every_declared_symbol_and_needs_no_hip_runtime():
    """include/scs_io.h <-> libscs_io.so <-> the ctypes table; the library must not depend on a HIP runtime (it is loaded before
    torch has brought its own)."""
    import subprocess
    from cnn_cort import _hip
    src = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'scs_io.h')).read(), flags=re.S)
    names = sorted(set(re.findall(r'\b(scs_[a-z_0-9]+)\s*\(', src)))
    lib = _hip.load_io_library()
    assert set(names) == set(_hip.IO_SIGNATURES) and all(hasattr(lib, n) for n in names)
    needed = subprocess.run(['readelf', '-d', _hip.IO_LIB_PATH], capture_output=True, text=True).stdout
    assert 'libamdhip64' not in needed and 'libscs.so' not in needed


def test_errors_are_reported_not_fatal():
    from cnn_cort import _hip
    lib = _hip.load_library()
    assert lib.scs_create(0, None) < 0
    assert b'NULL' in lib.scs_last_error() or b'null' in lib.scs_last_error()
    assert lib.scs_destroy(None) == 0
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(_hip.ScsError):
            _hip.Handle(0)


def test_python_surface_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a device is present')
    import numpy as np
    from cnn_cort.nets import Net
    net = Net()
    X = {'in1': np.zeros((1, 1, 32, 32), np.float32), 'in2': np.zeros((1, 1, 32, 32), np.float32),
         'in3': np.zeros((1, 1, 32, 32), np.float32), 'in4': np.zeros((1, 15), np.float32)}
    with pytest.raises(RuntimeError):
        net.predict(X)


def test_header_is_plain_c_and_links(tmp_path):
    """include/scs.h is C99 (no C++-isms) and a C program links against libscs.so with gcc alone (not run here)."""
    import subprocess
    from conftest import PKG, ROOT
    lib_dir = os.path.join(PKG, 'lib')
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'c', 'abi_host.c'), '-o', str(tmp_path / 'abi_host'),
                           '-L', lib_dir, '-lscs', '-Wl,-rpath,' + lib_dir, '-Wl,--allow-shlib-undefined'])
