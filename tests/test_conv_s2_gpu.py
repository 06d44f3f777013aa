"""conv3x3s2_f16x3_kernel (csrc/conv_fwd_s2.h): the 3x3 stride-2 convolutions of the discriminator's / encoders' down path
(conv2d_resample.py:119-122: blur to 2H + 1, then stride 2 without padding; the encoders' padded form) against torch in fp64;
which kernel ran is read back from ``pasta_conv2d_plan`` (kernel 10)."""

import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _kernel_id(n, ci, co, h, pad):
    from torch_utils import custom_ops
    oh = (h + 2 * pad - 3) // 2 + 1
    desc = custom_ops.ConvDesc(N=n, C_in=ci, H=h, W=h, C_out=co, OH=oh, OW=oh, kh=3, kw=3, stride=2, pad_h=pad, pad_w=pad, groups=1,
                               transposed=0, flip=0, math=0)
    k = ctypes.c_int()
    assert custom_ops.get_plugin().pasta_conv2d_plan(ctypes.byref(desc), 0, None, None, None, None, ctypes.byref(k)) == 0
    return k.value


@pytest.mark.parametrize('n,ci,co,h,pad,dedicated', [
    (1, 64, 64, 257, 0, True),          # one tile row of 128 outputs, 64-row tile (the discriminator's first down convolution, one sample)
    (3, 32, 128, 129, 0, True),         # two tile rows of 64, 128-row tile
    (9, 48, 96, 65, 0, True),           # four tile rows of 32; channel tails on both sides
    (40, 20, 40, 33, 0, True),          # eight tile rows of 16; 20 input channels: a chunk and a quarter
    (3, 32, 128, 128, 1, True),         # the encoders' padded form: the first row / column of every window lies outside
    (40, 64, 72, 32, 1, True),
    (2, 128, 256, 129, 0, True),        # 8192 pixels: the K-sliced small-plane path takes it...
    (40, 64, 64, 17, 0, False),         # 8 x 8 outputs: the base kernel
    (3, 8, 64, 129, 0, False),          # fewer than 16 input channels
])
def test_stride2_convolution(n, ci, co, h, pad, dedicated):
    from torch_utils.ops import conv2d_gradfix as cg
    oh = (h + 2 * pad - 3) // 2 + 1
    if n * oh * oh <= 8192:
        dedicated = False
    assert (_kernel_id(n, ci, co, h, pad) == 10) == dedicated
    g = torch.Generator().manual_seed(ci + co + h)
    x = torch.randn([n, ci, h, h], generator=g)
    w = torch.randn([co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)
    w[co // 2:] *= 1e-3
    b = torch.randn([co], generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=pad)
    y = cg.conv2d(x.cuda(), w.cuda(), stride=2, padding=pad)
    assert y.shape == ref.shape
    d = (y.double().cpu() - ref).abs()
    assert float(d.max() / ref.abs().max()) < 3e-6
    assert float((d.amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3))).max()) < 1e-5          # per output channel
    # with the fused epilogue of Conv2dLayer (bias, lrelu, gain, clamp) and the gradients of the layer
    xc = x.cuda().requires_grad_(True); wc = w.cuda().requires_grad_(True); bc = b.cuda().requires_grad_(True)
    z = cg.conv2d_bias_act(xc, wc, bc, stride=2, padding=pad, act='lrelu', gain=2 ** 0.5, clamp=256)
    dy = torch.randn(z.shape, generator=g)
    gx, gw, gb = torch.autograd.grad(z, [xc, wc, bc], dy.cuda())
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = torch.nn.functional.conv2d(x64, w64, stride=2, padding=pad) + b64.reshape(1, -1, 1, 1)
    z64 = (torch.where(z.detach().cpu() >= 0, pre, pre * 0.2) * 2 ** 0.5).clamp(-256, 256)       # the slopes the GPU took
    rx, rw, rb = torch.autograd.grad(z64, [x64, w64, b64], dy.double())
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())
    assert rel(z, z64) < 3e-6 and rel(gx, rx) < 5e-6 and rel(gw, rw) < 1e-5 and rel(gb, rb) < 1e-5
