"""3x3 stride-1 convolutions on the two-dimensional pixel tiles of conv_fwd_rows2d_bf16x6_kernel (csrc/conv_fwd_rows2d_bf16x6.h:
R output rows x 128/R columns per workgroup, the R + 2 input rows staged once per 16-channel chunk) against torch's CPU
convolution in fp64: forward, input gradient (flipped, transposed weights), groups, channel tails, K slices (few pixels),
two-row tiles, and the fused epilogue.  The planner's report says which kernel a shape takes."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # n, cin, cout, h, w, groups, expected kernel code (7: eight waves on 128 x 256 tiles; 4: four-row, 5: two-row tiles of the
           # 128 x 128 tile; 6: eight-row tiles of the 64 x 256 tile)
    (9, 128, 128, 32, 32, 1, 7),
    (3, 64, 256, 64, 64, 1, 7),
    (2, 72, 200, 64, 32, 1, 7),        # channel tails on both operands
    (3, 128, 256, 32, 64, 2, 7),       # two groups
    (2, 128, 128, 36, 64, 1, 4),       # 36 rows: no multiple of 8 -> four-row tiles on four waves
    (2, 128, 128, 66, 64, 1, 5),       # 66 rows: two-row tiles of 64 columns
    (1, 512, 512, 32, 32, 1, 7),       # 1024 pixels: the K range is sliced
    (2, 256, 128, 128, 128, 1, 7),
    (3, 64, 64, 64, 64, 1, 6),         # 64 x 256 tile: eight rows x 32 columns
    (1, 40, 48, 128, 96, 1, 6),
]


def _kernel_code(desc):
    from torch_utils.ops import _native
    kernel = ctypes.c_int()
    assert _native.lib().pasta_conv2d_plan(ctypes.byref(desc), 0, None, None, None, None, ctypes.byref(kernel)) == 0
    return kernel.value


@pytest.mark.parametrize('n,cin,cout,h,w,groups,code', CASES)
def test_forward_and_input_gradient_match_fp64(n, cin, cout, h, w, groups, code):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(h + cin)
    x = torch.randn([n, cin, h, w], generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn([cout, cin // groups, 3, 3], generator=g, dtype=torch.float64) / (9 * cin // groups) ** 0.5).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, wt, padding=1, groups=groups)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    rdx, rdw = torch.autograd.grad(y, [x, wt], dy)
    xg = x.detach().float().cuda().requires_grad_(True)
    wg = wt.detach().float().cuda().requires_grad_(True)
    yg = cg.conv2d(xg, wg, padding=1, groups=groups)
    dx, dw = torch.autograd.grad(yg, [xg, wg], dy.float().cuda())
    for got, ref in ((yg, y), (dx, rdx), (dw, rdw)):
        assert float((got.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max()) < 5e-6        # as tests/test_conv_precision_gpu.py: K up to 4608 fp32 accumulations
    if cg.conv_math in ('default', 'bf16x6') and os.environ.get('PASTA_ROWS2D', '8') == '8':
        cfg = cg._Cfg((False, 1, 1, 1, 0, 0, groups, 1.0))
        assert _kernel_code(cg._desc(cfg, x.shape, cout, h, w, 3, 3)) == code


def test_fused_epilogue_on_two_dimensional_tiles():
    """bias + lrelu + gain + clamp + residual in the epilogue (conv2d_bias_act), and the forward-only modulated form
    (demodulation + noise) -- its input scale keeps the row kernel, the epilogue operands are shared code."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(5)
    x = torch.randn([4, 128, 64, 64], generator=g)
    wt = torch.randn([128, 128, 3, 3], generator=g) / 34
    b = torch.randn([128], generator=g)
    res = torch.randn([4, 128, 64, 64], generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1) + res.double() + b.double()[None, :, None, None]
    ref = (torch.nn.functional.leaky_relu(ref, 0.2) * 1.3).clamp(-2.0, 2.0)
    y = cg.conv2d_bias_act(x.cuda(), wt.cuda(), b.cuda(), padding=1, act='lrelu', gain=1.3, clamp=2.0, residual=res.cuda())
    assert float((y.cpu().double() - ref).abs().max()) < 2e-5      # values up to 2, K = 1152 fp32 accumulations


@pytest.mark.parametrize('mode', ['0', '4'])
def test_other_kernels_still_serve_the_same_shapes(mode):
    """PASTA_ROWS2D=0: the one-dimensional row kernel (the fallback for planes the 2-D tiles do not divide); =4: the four-wave
    2-D tiles everywhere (what the reduced arithmetics, 16-bit storage and launches with an input scale run) -- on this file's
    and the precision file's shapes, in a child process."""
    import subprocess
    import sys
    if os.environ.get('PASTA_ROWS2D_CHILD'):
        pytest.skip('already the child')
    env = dict(os.environ, PASTA_ROWS2D=mode, PASTA_ROWS2D_CHILD='1')
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), os.path.join(here, 'test_conv_precision_gpu.py'), '-x', '-q',
                          '-k', 'not plan_reports and not still_serve and not every_tiling'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
