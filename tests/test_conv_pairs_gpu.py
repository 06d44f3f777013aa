"""Stride-2 3x3 conv_transpose2d on the parity-pair mode of the row-reuse kernel (csrc/conv_fwd_bf16x6.h, PAIR): every
upsampling layer (conv2d_resample.py:113-131 of the reference: transposed convolution, then the low-pass) and the input
gradient of every stride-2 convolution.  Checked against torch's CPU convolution in fp64 on shapes that take the pair
kernel with and without the remainder row / column, and through the planner's own report of which kernel runs."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # n, cin, cout, h, w, pad, output_padding   (lattices above 8192 pixels: below, the small-plane plans run)
    (9, 64, 64, 32, 32, 0, 0),       # 64 x 256 tile, OH = 2H + 1: remainder row and column
    (9, 128, 128, 32, 32, 0, 0),     # 128 x 128 tile
    (5, 64, 128, 32, 64, 1, 1),      # OH = 2H: no remainder; the odd column carries two taps
    (5, 32, 64, 64, 32, 1, 1),
    (9, 256, 128, 32, 32, 0, 0),
    (9, 48, 72, 32, 32, 0, 0),       # channel counts that are no multiple of the 16-channel chunk / the tile height
    (3, 64, 64, 64, 64, 0, 0),
    (1, 128, 64, 128, 128, 0, 0),    # the 128 -> 64 upsampling layer onto 257 x 257: pair kernel + remainder launch on the side stream
    (1, 64, 128, 128, 128, 0, 0),
]


def _plan_kernel(cfg_desc):
    from torch_utils.ops import _native
    kernel, launches = ctypes.c_int(), ctypes.c_int()
    assert _native.lib().pasta_conv2d_plan(ctypes.byref(cfg_desc), 0, None, None, None, ctypes.byref(launches), ctypes.byref(kernel)) == 0
    return kernel.value, launches.value


@pytest.mark.parametrize('math', ['bf16x6', 'default'])
@pytest.mark.parametrize('n,cin,cout,h,w,pad,opad', CASES)
def test_transposed_stride2_pairs_match_fp64(n, cin, cout, h, w, pad, opad, math):
    """Both fp32-class arithmetics (six-product split-bf16, and the default three products of scaled fp16 pieces) run these
    shapes on the pair kernel -- the default arithmetic on the one-pass kernel where it applies (round 5) -- against fp64."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(h * 7 + cin)
    x = torch.randn([n, cin, h, w], generator=g)
    wt = torch.randn([cin, cout, 3, 3], generator=g) / (cin * 9) ** 0.5
    ref = torch.nn.functional.conv_transpose2d(x.double(), wt.double(), stride=2, padding=pad, output_padding=opad)
    old, cg.conv_math = cg.conv_math, math
    try:
        y = cg.conv_transpose2d(x.cuda(), wt.cuda(), stride=2, padding=pad, output_padding=opad)
        cfg = cg._Cfg((True, 2, pad, pad, opad, opad, 1, 1.0))
        kernel, launches = _plan_kernel(cg._desc(cfg, x.shape, cout, ref.shape[2], ref.shape[3], 3, 3))
    finally:
        cg.conv_math = old
    assert y.shape == ref.shape
    err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    t2 = (math == 'default' and pad == 0 and ((h % 8 == 0 and w % 32 == 0) or (h % 16 == 0 and w % 16 == 0)) and cin >= 16 and os.environ.get('PASTA_CONV_T2') in (None, '2'))
    if t2:                          # round 5: the one-pass kernel (conv_fwd_t2.h), remainder row / column inside the launch
        assert kernel == 13 and launches == 1
    elif opad == 1:                 # even output planes: always the pair kernel
        assert kernel == 3 and launches == 1
    elif h * w >= 128 * 128 or os.environ.get('PASTA_T2_PAIR') == '2':     # with a remainder row / column: planes of 128 x 128 and larger (csrc/conv_igemm.hip, pair_launch_ok)
        assert kernel == 3 and launches == 2
    else:
        assert kernel == 1 and launches == 1        # the four parity classes share one grid of conv_fwd_bf16x6_kernel


def test_stride2_convolution_input_gradient_takes_the_pair_kernel():
    """dgrad of a 3x3 stride-2 convolution over a 65 x 65 plane (the blurred 64 x 64 of a discriminator block) = a
    transposed convolution from 32 x 32 onto 65 x 65."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(3)
    x = torch.randn([9, 64, 65, 65], generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn([128, 64, 3, 3], generator=g, dtype=torch.float64) / 24).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, wt, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    rdx, rdw = torch.autograd.grad(y, [x, wt], dy)
    xg = x.detach().float().cuda().requires_grad_(True)
    wg = wt.detach().float().cuda().requires_grad_(True)
    dx, dw = torch.autograd.grad(cg.conv2d(xg, wg, stride=2), [xg, wg], dy.float().cuda())
    assert float((dx.cpu().double() - rdx).abs().max() / rdx.abs().max()) < 2e-6
    assert float((dw.cpu().double() - rdw).abs().max() / rdw.abs().max()) < 2e-6


def test_upsampling_layer_through_conv2d_resample():
    """conv2d_resample(up=2) -- transposed convolution + low-pass -- against the CPU oracle (oracle/ref_ops.py, pinned by the
    reference's own fixtures), in its defining composition (fast=False: upsample, convolve, no transposed convolution)."""
    from oracle import ref_ops as R
    from torch_utils.ops import conv2d_resample, upfirdn2d
    g = torch.Generator().manual_seed(11)
    x = torch.randn([9, 64, 32, 32], generator=g)
    wt = torch.randn([64, 64, 3, 3], generator=g) / 24
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    for flip_weight in (False, True):
        y = conv2d_resample.conv2d_resample(x.cuda(), wt.cuda(), f=f.cuda(), up=2, padding=1, flip_weight=flip_weight)
        ref = R.conv2d_resample(x.double(), wt.double(), f=f.double(), up=2, padding=1, flip_weight=flip_weight, fast=False)
        assert y.shape == ref.shape
        assert float((y.cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-5


def test_small_planes_on_the_pair_kernel_with_remainder():
    """PASTA_T2_PAIR=2 sends every eligible shape to the pair kernel (the planner otherwise keeps planes below 128 x 128 with
    a remainder row / column on the per-class launch, which is faster there): the same cases, in a child process."""
    import os
    import subprocess
    import sys
    if os.environ.get('PASTA_PAIRS_CHILD'):
        pytest.skip('already the child')
    env = dict(os.environ, PASTA_T2_PAIR='2', PASTA_PAIRS_CHILD='1')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k', 'fp64 or gradient or resample'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
