"""The C ABI: every symbol include/pasta_hip.h declares is exported by the built library and typed in the loader.
No device work is launched here (host-only entry points may be called)."""

import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'pasta_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pasta_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_surface():
    syms = _declared_symbols()
    for must in ['pasta_upfirdn2d', 'pasta_bias_act', 'pasta_conv2d', 'pasta_conv2d_wgrad', 'pasta_spade_norm', 'pasta_last_error']:
        assert must in syms


def test_library_exports_every_declared_symbol():
    from torch_utils import custom_ops
    path = custom_ops.build()
    lib = ctypes.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(lib, name), f'{name} declared in pasta_hip.h but not exported'
    assert set(custom_ops.ABI) == set(_declared_symbols()), 'loader table and header disagree'


def test_host_only_entry_points():
    from torch_utils import custom_ops
    lib = custom_ops.get_plugin()
    assert lib.pasta_abi_version() == custom_ops.EXPECTED_ABI == 21
    assert b'gfx950' in lib.pasta_build_info()
    d = custom_ops.ConvDesc(N=2, C_in=8, H=16, W=16, C_out=8, OH=16, OW=16, kh=3, kw=3, stride=1, pad_h=1, pad_w=1, groups=1, transposed=0, flip=0, math=0)
    # 2 x 256 partial operand maxima + 32 row scales of the packed weights (PASTA_MATH_F16X3) + [taps][I_pad8][O_pad32] packed weights, 6 B each
    assert lib.pasta_conv2d_workspace(ctypes.byref(d)) == 2 * 256 * 4 + 32 * 4 + 9 * 8 * 32 * 6
    assert lib.pasta_conv2d_wgrad_workspace(ctypes.byref(d)) > 0
    assert lib.pasta_conv2d_tile(ctypes.byref(d)) == 2
    bad = custom_ops.ConvDesc(N=2, C_in=8, H=16, W=16, C_out=8, OH=15, OW=16, kh=3, kw=3, stride=1, pad_h=1, pad_w=1, groups=1, transposed=0, flip=0, math=0)
    assert lib.pasta_conv2d_workspace(ctypes.byref(bad)) == -1
    assert b'conv2d output is 16x16' in lib.pasta_last_error()
    with pytest.raises(RuntimeError):
        custom_ops.check(lib, 1)
    assert lib.pasta_bias_grad_workspace(16 * 64 * 32 * 32, 64, 32 * 32) == 64 * 16 * 4


def test_loader_refuses_a_library_of_another_abi(monkeypatch):
    """the ctypes structure layouts belong to one ABI revision; PASTA_LIB_AB (same-box A/B of two builds) must not slip a stale
    library past them."""
    from torch_utils import custom_ops
    monkeypatch.setattr(custom_ops, '_cached_plugins', {})
    monkeypatch.setenv('PASTA_LIB_AB', os.path.join(ROOT, 'no_such_library.so'))
    with pytest.raises(RuntimeError, match='does not exist'):
        custom_ops.get_plugin()
    monkeypatch.delenv('PASTA_LIB_AB')
    monkeypatch.setattr(custom_ops, 'EXPECTED_ABI', custom_ops.EXPECTED_ABI - 1)       # as if the tree were one revision behind the library
    with pytest.raises(RuntimeError, match='reports ABI'):
        custom_ops.get_plugin()


def test_convolution_family_is_built_from_parallel_units():
    """the convolution kernels live in translation units of their own (csrc/conv_tu_*.hip behind csrc/conv_launch.h), so a
    forced build is a minute on eight cores; the planner unit instantiates no kernel template of the family."""
    from torch_utils import custom_ops
    units = [os.path.basename(s) for s in custom_ops._sources()]
    assert sum(u.startswith('conv_tu_') for u in units) >= 12 and 'conv_igemm.hip' in units
    text = open(os.path.join(ROOT, 'pasta-gan_amd', 'csrc', 'conv_igemm.hip')).read()
    for kernel in ('conv_fwd_rows2d_bf16x6_kernel<', 'conv_fwd_bf16x6_kernel<', 'conv_wgrad3x3_bf16x6_kernel<', 'conv_fwd_kernel<'):
        assert kernel not in text, kernel
