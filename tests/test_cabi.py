"""CPU: the C-ABI library loads and exports every symbol include/remixt_amd.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from remixt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'remixt_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rmx_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), 'missing export: ' + n


def test_library_exports_nothing_undeclared():
    """Every rmx_* entry point the library exports is declared (and documented) in the header: no debugging hooks ship."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith('rmx_') and ' T ' in line))
    assert exported == declared_symbols()


def test_binding_table_matches_header():
    assert sorted(_lib.SYMBOLS) == declared_symbols()
    lib = _lib.load()
    assert lib.rmx_num_kernels() > 10
    assert lib.rmx_kernel_name(3).decode() == 'k_fb'


def test_struct_layout():
    # struct rmx_problem: 8 int32 + 9 pointers + 1 double
    assert ctypes.sizeof(_lib.RmxProblem) == 8 * 4 + 9 * 8 + 8


def test_no_device_is_an_error_not_a_fallback():
    """Without a HIP device the create call must fail (RMX_EDEVICE), never compute on the CPU."""
    import numpy as np
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from remixt_amd import bpmodel
    cn = np.ones((3, 2, 2, 2), dtype=np.int64)
    with pytest.raises(RuntimeError):
        bpmodel.RemixtModel(2, 3, 0, True, cn, np.zeros((1, 2), dtype=np.int64), np.array([0.1, 0.1]), np.ones(3), np.ones(3),
                            np.ones((3, 2)), np.array([0, 0, 1]), -np.ones(3, dtype=np.int64), np.zeros(3, dtype=np.int64), 10., 1e-6)
