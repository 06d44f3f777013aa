"""16-bit activation storage in the convolution family (include/pasta_hip.h, pasta_conv_desc.io_dtype): fp16 / bf16 tensors
in HBM, the stored element is the matrix-core operand (one product per multiply-add), fp32 accumulation and epilogue, one
rounding on the way out.  Yardstick: the fp32-equivalent path of this package on the SAME 16-bit-valued operands -- the
products are then identical (exact in fp32 either way), so results agree to the output's rounding (forward, input gradient)
or to fp32 summation order (weight gradient, which stays fp32)."""

import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}

CASES = [   # name, x shape, w shape, kwargs of conv2d / conv_transpose2d, transposed
    ('rows_3x3_128',        [4, 128, 64, 64],  [128, 128, 3, 3], dict(padding=1), False),          # row-reuse kernel, 128x128 tile
    ('rows_3x3_64x256',     [2, 64, 128, 128], [64, 64, 3, 3],   dict(padding=1), False),          # row-reuse kernel, 64x256 tile
    ('stride2_3x3',         [4, 64, 65, 65],   [128, 64, 3, 3],  dict(stride=2), False),           # base kernel, stride-2 weight gradient
    ('transposed_s2',       [8, 128, 32, 32],  [128, 64, 3, 3],  dict(stride=2), True),            # four parity classes in one grid
    ('pointwise',           [4, 192, 64, 64],  [128, 192, 1, 1], dict(), False),                   # 1x1 weight gradient kernel
    ('small_plane_splitk',  [8, 512, 8, 8],    [512, 512, 3, 3], dict(padding=1), False),          # K sliced over workgroups + reduce kernel
    ('grouped',             [1, 4 * 64, 128, 128], [4 * 64, 64, 3, 3], dict(padding=1, groups=4), False),
]


def _run(x, w, kw, transposed):
    from torch_utils.ops import conv2d_gradfix as cg
    return (cg.conv_transpose2d if transposed else cg.conv2d)(x, w, **kw)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
@pytest.mark.parametrize('name,xs,ws,kw,transposed', CASES, ids=[c[0] for c in CASES])
def test_native_16bit_conv_matches_the_fp32_path_on_the_same_operands(name, xs, ws, kw, transposed, dtype):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(len(name))
    fan = ws[1] * ws[2] * ws[3] if not transposed else ws[0] * ws[2] * ws[3] / 4
    x16 = torch.randn(xs, generator=g).to(dtype).cuda().requires_grad_(True)
    w = (torch.randn(ws, generator=g) / np.sqrt(fan)).to(dtype).float().cuda().requires_grad_(True)    # weights on the 16-bit grid, fp32 tensor
    y16 = _run(x16, w, kw, transposed)
    assert y16.dtype == dtype
    # this launch really was native: the planner accepts the descriptor
    cfg = cg._Cfg((transposed, kw.get('stride', 1), kw.get('padding', 0), kw.get('padding', 0), 0, 0, kw.get('groups', 1), 1.0))
    c_out = ws[1] * kw.get('groups', 1) if transposed else ws[0]
    assert cg._native16('conv', cg._desc(cfg, xs, c_out, y16.shape[2], y16.shape[3], ws[2], ws[3], dtype))
    x32 = x16.detach().float().requires_grad_(True)
    w32 = w.detach().clone().requires_grad_(True)
    y32 = _run(x32, w32, kw, transposed)
    scale = float(y32.abs().max())
    assert float((y16.float() - y32).abs().max()) <= 1.01 * EPS[dtype] * scale          # one rounding of the output
    dy16 = torch.randn(y16.shape, generator=g).to(dtype).cuda()
    dx16, dw16 = torch.autograd.grad(y16, [x16, w], dy16)
    dx32, dw32 = torch.autograd.grad(y32, [x32, w32], dy16.float())
    assert dx16.dtype == dtype and dw16.dtype == torch.float32
    assert float((dx16.float() - dx32).abs().max()) <= 1.01 * EPS[dtype] * float(dx32.abs().max())
    assert float((dw16 - dw32).abs().max()) <= 2e-5 * float(dw32.abs().max())          # same products, fp32 sums in another order


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
def test_fused_epilogue_and_residual_in_16bit(dtype):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(7)
    x = torch.randn([4, 64, 64, 64], generator=g).to(dtype).cuda()
    w = (torch.randn([128, 64, 3, 3], generator=g) / 24).to(dtype).float().cuda()
    b = (torch.randn([128], generator=g) * 0.1).to(dtype).cuda()
    r = torch.randn([4, 128, 64, 64], generator=g).to(dtype).cuda()
    y = cg.conv2d_bias_act(x, w, b, padding=1, act='lrelu', gain=np.sqrt(2), clamp=3.0, residual=r)
    ref = cg.conv2d_bias_act(x.float(), w, b.float(), padding=1, act='lrelu', gain=np.sqrt(2), clamp=3.0, residual=r.float())
    assert y.dtype == dtype and float((y.float() - ref).abs().max()) <= 1.01 * EPS[dtype] * float(ref.abs().max())
    assert float(y.float().abs().max()) <= 3.0 + 1e-6


def test_few_channel_layers_fall_back_to_an_fp32_launch():
    """RGB stems and ToRGB heads have no 16-bit kernel: the planner says so and the launch converts (same result type)."""
    from torch_utils.ops import conv2d_gradfix as cg
    x = torch.randn([2, 3, 64, 64]).half().cuda()
    w = torch.randn([64, 3, 1, 1]).cuda()
    cfg = cg._Cfg((False, 1, 0, 0, 0, 0, 1, 1.0))
    assert not cg._native16('conv', cg._desc(cfg, x.shape, 64, 64, 64, 1, 1, torch.float16))
    y = cg.conv2d(x, w)
    assert y.dtype == torch.float16
    ref = cg.conv2d(x.float(), w)
    assert float((y.float() - ref).abs().max()) <= 1.01 * 2.0 ** -11 * float(ref.abs().max())
