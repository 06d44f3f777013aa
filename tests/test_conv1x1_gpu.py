"""conv1x1_f16x3_kernel (csrc/conv_fwd_1x1.h): pointwise convolutions and their input gradients under the default arithmetic, and the
two-tensor form that serves ``Conv2dLayer(cat([x, side], 1))`` of the synthesis blocks (networks.py:5698-5700) without the
concatenation.  Against torch's fp64 convolution; which kernel ran is read back from ``pasta_conv2d_plan``."""

import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _kernel_id(n, ci, co, h, w, flags=0, transposed=0):
    from torch_utils import custom_ops
    desc = custom_ops.ConvDesc(N=n, C_in=ci, H=h, W=w, C_out=co, OH=h, OW=w, kh=1, kw=1, stride=1, pad_h=0, pad_w=0, groups=1,
                               transposed=transposed, flip=0, math=0)
    k = ctypes.c_int()
    assert custom_ops.get_plugin().pasta_conv2d_plan(ctypes.byref(desc), flags, None, None, None, None, ctypes.byref(k)) == 0
    return k.value


@pytest.mark.parametrize('n,ci,co,h,w,pointwise', [
    (3, 64, 64, 64, 64, True),          # 64-row tile, 256-pixel tiles, two rounds  (more than 8192 pixels in all: below that the K-sliced small-plane path runs)
    (9, 128, 64, 32, 32, True),
    (40, 48, 64, 16, 16, True),         # one and a half rounds: the second chunk of the last round meets zero activations
    (40, 20, 40, 16, 16, True),         # channel tail inside a chunk, output rows beyond C_out in the tile
    (36, 192, 128, 16, 16, True),       # 128-row tile, 128-pixel tiles
    (70, 320, 256, 16, 8, True),        # two output-channel tiles
    (9, 576, 512, 32, 32, True),
    (40, 64, 64, 15, 15, False),        # planes that do not divide into tiles: the one-tap path of the base kernels
    (9, 8, 64, 32, 32, False),          # fewer than 16 input channels
    (2, 64, 64, 32, 32, False),         # 2048 pixels: the small-plane path
])
def test_pointwise_convolution_and_gradients(n, ci, co, h, w, pointwise):
    from torch_utils.ops import conv2d_gradfix as cg
    assert (_kernel_id(n, ci, co, h, w) == 9) == pointwise
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn([n, ci, h, w], generator=g)
    wt = torch.randn([co, ci, 1, 1], generator=g) / ci ** 0.5
    wt[co // 2:] *= 1e-3                                     # rows of very different size: one scale per row
    dy = torch.randn([n, co, h, w], generator=g)
    x64 = x.double().requires_grad_(True); w64 = wt.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64)
    rx, rw = torch.autograd.grad(y64, [x64, w64], dy.double())
    xc = x.cuda().requires_grad_(True); wc = wt.cuda().requires_grad_(True)
    y = cg.conv2d(xc, wc)
    gx, gw = torch.autograd.grad(y, [xc, wc], dy.cuda())
    assert _rel(y, y64) < 3e-6 and _rel(gx, rx) < 3e-6 and _rel(gw, rw) < 1e-5
    per_row = ((y.detach().double().cpu() - y64.detach()).abs().amax(dim=(0, 2, 3)) / y64.detach().abs().amax(dim=(0, 2, 3)))
    assert float(per_row.max()) < 1e-5                      # the quiet output channels as accurate as the loud ones
    # the transposed operator with the same weight (what the input gradient launches): also the pointwise kernel
    if pointwise and co >= 16 and ci > 32:
        assert _kernel_id(n, co, ci, h, w, transposed=1) == 9


@pytest.mark.parametrize('act,clamp', [('linear', 256.0), ('lrelu', None)])
@pytest.mark.parametrize('n,c1,c2,co,hw', [(9, 64, 64, 64, 32), (33, 128, 64, 128, 16), (33, 256, 64, 256, 16), (33, 20, 44, 64, 16), (40, 64, 64, 64, 15)])
def test_two_tensor_form_equals_the_concatenation(n, c1, c2, co, hw, act, clamp):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(c1 + c2 + co)
    x1 = torch.randn([n, c1, hw, hw], generator=g) * 3
    x2 = torch.randn([n, c2, hw, hw], generator=g) * 0.05      # the second operand much quieter: one scale serves both
    wt = torch.randn([co, c1 + c2, 1, 1], generator=g)
    b = torch.randn([co], generator=g)
    dy = torch.randn([n, co, hw, hw], generator=g)
    gain = 0.125
    a, c, w_, b_ = (t.cuda().requires_grad_(True) for t in (x1, x2, wt, b))
    fused = cg.cat1x1_available(a, c, w_)
    assert fused == (hw * hw % 128 == 0)
    y = cg.conv2d_cat1x1_bias_act(a, c, w_, b_, act=act, clamp=clamp, wgain=gain)
    got = (y,) + torch.autograd.grad(y, [a, c, w_, b_], dy.cuda())
    def ref():
        a, c, w_, b_ = x1.double().requires_grad_(True), x2.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
        z = torch.nn.functional.conv2d(torch.cat([a, c], 1), w_ * gain) + b_.reshape(1, -1, 1, 1)
        if act == 'lrelu':
            # the slope each element took on the GPU (a pre-activation within rounding of zero may take the other one in fp64: that is
            # not what this test is about)
            z = torch.where(y.detach().cpu() >= 0, z, z * 0.2) * 2 ** 0.5
        if clamp is not None:
            z = z.clamp(-clamp, clamp)
        return (z,) + torch.autograd.grad(z, [a, c, w_, b_], dy.double())
    want = ref()
    for name, u, v in zip(['y', 'dx1', 'dx2', 'dw', 'db'], got, want):
        assert u.shape == v.shape and (u.is_contiguous() or not fused), name
        assert _rel(u, v) < (3e-6 if name != 'dw' else 1e-5), (name, _rel(u, v))


def cg_available(x, side, w):
    from torch_utils.ops import conv2d_gradfix as cg
    return cg.cat1x1_available(x, side, w)


def test_merge_layer_of_a_synthesis_block_uses_it():
    """_merge_without_cat against the layer on the concatenated tensor (the reference's expression), values and all gradients."""
    from training import networks
    g = torch.Generator().manual_seed(5)
    layer = networks.Conv2dLayer(128 + 64, 128, kernel_size=1, conv_clamp=256).cuda()
    with torch.no_grad():
        layer.bias.copy_(torch.randn([128], generator=g))
    x = torch.randn([9, 128, 32, 32], generator=g).cuda().requires_grad_(True)
    side = torch.randn([9, 64, 32, 32], generator=g).cuda().requires_grad_(True)
    dy = torch.randn([9, 128, 32, 32], generator=g).cuda()
    assert cg_available(x, side, layer.weight)
    y0 = layer(torch.cat([x, side], dim=1))
    g0 = torch.autograd.grad(y0, [x, side, layer.weight, layer.bias], dy)
    y1 = networks._merge_without_cat(layer, x, side)
    g1 = torch.autograd.grad(y1, [x, side, layer.weight, layer.bias], dy)
    assert _rel(y1, y0) < 2e-6
    for u, v in zip(g1, g0):
        assert _rel(u, v) < 1e-5


@pytest.mark.parametrize('n,ci,co,h,w', [(5, 3, 64, 32, 32), (2, 6, 64, 64, 48), (3, 8, 128, 16, 16), (2, 1, 9, 8, 12), (4, 3, 70, 10, 6), (2, 3, 64, 15, 15)])
def test_few_channel_pointwise_weight_gradient(n, ci, co, h, w):
    """wgrad1x1_fewcin_kernel (the discriminator's fromrgb, the 6-channel pose stem): against fp64; planes that are not a multiple of four
    pixels stay on the (channel, tap)-pair kernel."""
    from torch_utils import custom_ops
    from torch_utils.ops import conv2d_gradfix as cg
    desc = custom_ops.ConvDesc(N=n, C_in=ci, H=h, W=w, C_out=co, OH=h, OW=w, kh=1, kw=1, stride=1, pad_h=0, pad_w=0, groups=1, transposed=0, flip=0, math=0)
    k = ctypes.c_int()
    assert custom_ops.get_plugin().pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(k)) == 0
    assert (k.value == 5) == ((h * w) % 4 == 0)
    g = torch.Generator().manual_seed(n + ci + co)
    x = torch.randn([n, ci, h, w], generator=g)
    wt = torch.randn([co, ci, 1, 1], generator=g)
    dy = torch.randn([n, co, h, w], generator=g)
    w64 = wt.double().requires_grad_(True)
    rw, = torch.autograd.grad(torch.nn.functional.conv2d(x.double(), w64), w64, dy.double())
    wc = wt.cuda().requires_grad_(True)
    gw, = torch.autograd.grad(cg.conv2d(x.cuda(), wc, wgain=0.25), wc, dy.cuda())
    assert gw.shape == rw.shape and _rel(gw, rw * 0.25) < 2e-6
