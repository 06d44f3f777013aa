"""Forward convolutions with fewer than 16 input channels and at least 64 (channel, tap) pairs -- the 7x7 RGB stems of the encoders
(networks.py:560-579, 4836-4883) -- on the matrix-core kernels in their packed-K mode (csrc/conv_fwd_bf16x6.h, KT): K runs over the
pairs, their offsets come from a table, the input is read from a zero-padded copy.  Against torch's CPU convolution in fp64, and the
planner's own report of the kernel."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # n, cin, cout, h, w, k, pad, stride
    (2, 3, 64, 128, 128, 7, 3, 1),       # the stem: 147 pairs, 64 x 256 tile
    (1, 3, 128, 128, 160, 7, 3, 1),      # 128 x 128 tile, non-square plane
    (2, 6, 64, 96, 96, 5, 2, 1),         # 150 pairs
    (2, 3, 64, 130, 130, 7, 0, 1),       # no padding: the input itself is read
    (2, 3, 64, 200, 200, 7, 3, 2),       # stride 2
    (3, 12, 96, 100, 100, 3, 1, 1),      # 108 pairs, a channel count that is no multiple of the tile height
]


@pytest.mark.parametrize('math', ['default', 'bf16x6'])
@pytest.mark.parametrize('n,cin,cout,h,w,k,pad,stride', CASES)
def test_few_channel_convolution_matches_fp64(n, cin, cout, h, w, k, pad, stride, math):
    from torch_utils.ops import conv2d_gradfix as cg, _native
    g = torch.Generator().manual_seed(h + 31 * cin + k)
    x = torch.randn([n, cin, h, w], generator=g)
    wt = torch.randn([cout, cin, k, k], generator=g) / (cin * k * k) ** 0.5
    b = torch.randn([cout], generator=g)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad), 0.2) * 2 ** 0.5
    old, cg.conv_math = cg.conv_math, math
    try:
        y = cg.conv2d_bias_act(x.cuda(), wt.cuda(), b.cuda(), stride=stride, padding=pad, act='lrelu')
        plain = cg.conv2d(x.cuda(), wt.cuda(), stride=stride, padding=pad)
        cfg = cg._Cfg((False, stride, pad, pad, 0, 0, 1, 1.0))
        desc = cg._desc(cfg, x.shape, cout, ref.shape[2], ref.shape[3], k, k)
        kernel, mth = ctypes.c_int(), ctypes.c_int()
        assert _native.lib().pasta_conv2d_plan(ctypes.byref(desc), 0, None, None, ctypes.byref(mth), None, ctypes.byref(kernel)) == 0
    finally:
        cg.conv_math = old
    assert kernel.value == 8, kernel.value                       # the packed-K mode
    assert mth.value == cg.MATH_CODES['f16x3' if math == 'default' else 'bf16x6']
    assert y.shape == ref.shape
    err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    ref_plain = torch.nn.functional.conv2d(x.double(), wt.double(), stride=stride, padding=pad)
    assert float((plain.cpu().double() - ref_plain).abs().max() / ref_plain.abs().max()) < 2e-6


def test_flipped_weight_and_gradients():
    """conv2d_resample(flip_weight=False) -- a true convolution -- takes the mode with mirrored offsets; the weight gradient of
    the layer (its own kernels) and the input gradient (a transposed convolution onto three channels) are unaffected."""
    from torch_utils.ops import conv2d_resample
    g = torch.Generator().manual_seed(5)
    x = torch.randn([2, 3, 128, 128], generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn([64, 3, 7, 7], generator=g, dtype=torch.float64) / 12).requires_grad_(True)
    for flip_weight in (True, False):
        w_eff = wt if flip_weight else wt.flip([2, 3])
        ref = torch.nn.functional.conv2d(x, w_eff, padding=3)
        dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
        rdx, rdw = torch.autograd.grad(ref, [x, wt], dy)
        xg = x.detach().float().cuda().requires_grad_(True)
        wg = wt.detach().float().cuda().requires_grad_(True)
        y = conv2d_resample.conv2d_resample(xg, wg, padding=3, flip_weight=flip_weight)
        assert float((y.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max()) < 2e-6, flip_weight
        dx, dw = torch.autograd.grad(y, [xg, wg], dy.float().cuda())
        assert float((dx.cpu().double() - rdx).abs().max() / rdx.abs().max()) < 1e-5
        assert float((dw.cpu().double() - rdw).abs().max() / rdw.abs().max()) < 1e-5


def test_small_tap_counts_stay_off_the_packed_k_mode():
    """3x3 over three channels (27 pairs): the launch is its output store, and the fp32 tile kernel's is the faster one (kernel 0); 1x1 over three
    channels: since round 5 the streaming few-channel kernel (kernel 11, tests/test_conv_fewch_gpu.py) -- never the packed-K mode (8)."""
    from torch_utils.ops import conv2d_gradfix as cg, _native
    for k, want in ((3, 0), (1, 11)):
        cfg = cg._Cfg((False, 1, k // 2, k // 2, 0, 0, 1, 1.0))
        desc = cg._desc(cfg, (4, 3, 128, 128), 64, 128, 128, k, k)
        kernel = ctypes.c_int()
        assert _native.lib().pasta_conv2d_plan(ctypes.byref(desc), 0, None, None, None, None, ctypes.byref(kernel)) == 0
        assert kernel.value == want, (k, kernel.value)
