"""Pointwise convolutions with very few channels on one side (round 5, csrc/conv_fwd_fewch.h: pasta_conv2d_plan kernels 11 / 12) -- the
RGB / pose stems, the ToRGB / parsing heads (reference networks.py:319-334, 5582-5611: 1x1 modulated convolutions without demodulation) and
their input gradients -- as streaming fp32 kernels on the raw weights: against fp64, with every epilogue the layers use."""

import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _kernel(n, ci, co, h, transposed=False, flags=0):
    from torch_utils import custom_ops
    from torch_utils.ops import _native
    d = custom_ops.ConvDesc(N=n, C_in=ci, H=h, W=h, C_out=co, OH=h, OW=h, kh=1, kw=1, stride=1, pad_h=0, pad_w=0, groups=1, transposed=int(transposed), flip=0, math=0)
    k, m = ctypes.c_int(-1), ctypes.c_int(-1)
    assert _native.lib().pasta_conv2d_plan(ctypes.byref(d), flags, None, None, ctypes.byref(m), None, ctypes.byref(k)) == 0
    return k.value, m.value


@pytest.mark.parametrize('n,ci,co,h,kind', [
    (6, 3, 64, 64, 11),          # fromrgb
    (3, 6, 64, 64, 11),          # the pose stem
    (4, 16, 128, 48, 11),        # sixteen input channels (the kernel's limit), a plane that is not a power of two
    (3, 9, 64, 64, 11),          # input gradient of a nine-channel head
    (4, 64, 3, 64, 12),          # ToRGB
    (3, 512, 9, 64, 12),         # image + parsing heads from 512 channels
    (5, 100, 6, 44, 12),         # a channel count that is not a multiple of eight (the tail loop)
    (2, 64, 3, 64, 0),           # 8192 pixels: stays on the tile kernels
    (4, 17, 17, 64, 0),          # more than sixteen channels on both sides
])
def test_few_channel_pointwise_convolutions(n, ci, co, h, kind):
    from torch_utils.ops import conv2d_gradfix as cg
    k, math = _kernel(n, ci, co, h)
    assert (k == kind) if kind else (k not in (11, 12))
    if kind:
        assert math == 1                                          # PASTA_MATH_F32: plain fp32 FMAs, nothing to scan
    g = torch.Generator().manual_seed(n + ci + co)
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    w = (torch.randn([co, ci, 1, 1], generator=g) / ci ** 0.5).cuda()
    b = torch.randn([co], generator=g).cuda()
    res = torch.randn([n, co, h, h], generator=g).cuda()
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu())
    y = cg.conv2d(x, w)
    assert _rel(y, ref) < 2e-6
    # wgain, residual, bias, lrelu, gain, clamp in the launch
    z = cg.conv2d_bias_act(x, w, b, act='lrelu', gain=1.5, clamp=1.0, wgain=0.7, residual=res)
    pre = ref * 0.7 + res.double().cpu() + b.double().cpu().reshape(1, -1, 1, 1)
    z64 = (torch.where(z.cpu() > 0, pre, pre * 0.2) * 1.5).clamp(-1.0, 1.0)
    assert _rel(z, z64) < 3e-6
    # the transposed operator (input gradients): weight [C_in, C_out, 1, 1]
    kt, _ = _kernel(n, ci, co, h, transposed=True)
    assert (kt == kind) if kind else True
    wt = (torch.randn([ci, co, 1, 1], generator=g) / ci ** 0.5).cuda()
    yt = cg.conv_transpose2d(x, wt)
    assert _rel(yt, torch.nn.functional.conv_transpose2d(x.double().cpu(), wt.double().cpu())) < 2e-6
    # gradients through the layer (input gradient = the other kind of kernel; weight gradient: the few-channel weight-gradient kernels)
    xs, ws = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    dy = torch.randn([n, co, h, h], generator=g).cuda()
    gx, gw = torch.autograd.grad(cg.conv2d(xs, ws), [xs, ws], dy)
    x64, w64 = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    rx, rw = torch.autograd.grad(torch.nn.functional.conv2d(x64, w64), [x64, w64], dy.double().cpu())
    assert _rel(gx, rx) < 3e-6 and _rel(gw, rw) < 2e-5


def test_modulated_head_takes_its_styles_in_the_weights():
    """ToRGB on the training path: conv(x, w, iscale = styles) (no demodulation) -- the few-output kernel multiplies the styles into its LDS copy
    of the weights (a workgroup lies inside one sample)."""
    from torch_utils.ops import conv2d_gradfix as cg
    assert _kernel(4, 64, 3, 64, flags=1)[0] == 12 and _kernel(4, 3, 64, 64, flags=1)[0] != 11      # an input scale: the few-output side only
    g = torch.Generator().manual_seed(3)
    n, ci, co, h = 4, 64, 3, 64
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    w = (torch.randn([co, ci, 1, 1], generator=g) / 8).cuda()
    s = (torch.randn([n, ci], generator=g) + 1).cuda()
    cfg = cg._Cfg((False, 1, 0, 0, 0, 0, 1, 0.125))
    y = cg._launch_conv(x, w, cfg, iscale=s)
    ref = torch.nn.functional.conv2d((x * s[:, :, None, None]).double().cpu(), w.double().cpu() * 0.125)
    assert _rel(y, ref) < 2e-6
    # a maxima row for the next convolution
    b = torch.randn([co], generator=g).cuda()
    z = cg.conv2d_bias_act(x, w, b, act='linear', clamp=256)
    hit = getattr(z, '_pasta_amax', None)
    if hit is not None:
        torch.cuda.synchronize()
        assert abs(float(hit[2].max()) - float(z.abs().max())) <= 1e-6 * float(z.abs().max())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,ci,co,h', [(6, 3, 64, 64), (4, 64, 3, 64), (3, 512, 9, 64)])
def test_few_channel_layers_stay_in_16_bit_storage(n, ci, co, h, dtype):
    """BASELINE config 5: with 16-bit activation storage the few-channel pointwise layers used to be converted to fp32 for the launch and back
    (no 16-bit tile kernel for them); the streaming kernels read and write the stored type (fp32 FMAs in between, one rounding on the way out)."""
    from torch_utils import custom_ops
    from torch_utils.ops import conv2d_gradfix as cg, _native
    io = cg.IO_CODES[dtype]
    d = custom_ops.ConvDesc(N=n, C_in=ci, H=h, W=h, C_out=co, OH=h, OW=h, kh=1, kw=1, stride=1, pad_h=0, pad_w=0, groups=1, transposed=0, flip=0, math=0, io_dtype=io)
    k = ctypes.c_int(-1)
    assert _native.lib().pasta_conv2d_plan(ctypes.byref(d), 4, None, None, None, None, ctypes.byref(k)) == 0 and k.value in (11, 12)
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn([n, ci, h, h], generator=g).cuda().to(dtype)
    w = (torch.randn([co, ci, 1, 1], generator=g) / ci ** 0.5).cuda()
    b = torch.randn([co], generator=g).cuda()
    y = cg.conv2d_bias_act(x, w, b, act='lrelu', gain=2 ** 0.5, clamp=256)
    assert y.dtype == dtype
    pre = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu()) + b.double().cpu().reshape(1, -1, 1, 1)
    ref = (torch.where(pre > 0, pre, pre * 0.2) * 2 ** 0.5).clamp(-256, 256)
    tol = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10         # one rounding of the result to the stored type
    err = (y.double().cpu() - ref).abs()
    assert float((err / (ref.abs() + 1e-2 * ref.abs().max())).max()) < tol
