"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/mxf_gp.h declares; the
product package never imports the oracle; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'mxf_gp.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mxf_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from mxfusion_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 20
    assert syms == _lib.ALL_SYMBOLS, (set(syms) ^ set(_lib.ALL_SYMBOLS))
    lib = ctypes.CDLL(_lib.LIB_PATH)          # raw dlopen: the symbols really are in the .so
    for s in syms:
        assert getattr(lib, s, None) is not None, s
    assert _lib.load().mxf_version() >= 100


def test_header_cites_reference_for_every_entry_point():
    txt = open(os.path.join(ROOT, 'include', 'mxf_gp.h')).read()
    for key in ('kernels/stationary.py:74-107', 'gp_regression.py:42-76', 'svgp_regression.py:43-109', 'normal.py:52-70',
                'var_trans.py:63-91', 'batch_loop.py:46-60'):
        assert key in txt, key


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'mxfusion_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dp, f)
                assert 'gp_oracle' not in src, os.path.join(dp, f)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_no_cpu_fallback():
    from mxfusion_amd import ops, _lib
    X = torch.rand(1, 4, 2, dtype=torch.float64)
    with pytest.raises(_lib.MXFError):
        ops.gram('rbf', X, None, torch.ones(1, 2, dtype=torch.float64), torch.ones(1, 1, dtype=torch.float64), True)
    h = ctypes.c_void_p()
    assert _lib.load().mxf_create(0, ctypes.byref(h)) != 0      # no device -> creation fails, nothing silently runs on CPU
