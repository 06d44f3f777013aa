"""3x3 stride-2 conv_transpose2d in one pass over the input lattice (round 5, csrc/conv_fwd_t2.h: pasta_conv2d_plan kernel 13) -- the transposed
convolution of every upsampling layer (reference conv2d_resample.py:104-115) and the input gradient of every stride-2 convolution: against fp64,
with the remainder row / column (OH = 2 H + 1) as edge tiles of the same launch, ragged batches and channel counts, groups, the modulated
layers' input scale, and through the operator API with its derivatives."""

import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _plan(n, ci, h, co, groups=1, pad=0, flags=0, w=None):
    from torch_utils import custom_ops
    from torch_utils.ops import _native
    w = h if w is None else w
    oh, ow = 2 * h + 1 - 2 * pad, 2 * w + 1 - 2 * pad
    d = custom_ops.ConvDesc(N=n, C_in=ci, H=h, W=w, C_out=co, OH=oh, OW=ow, kh=3, kw=3, stride=2, pad_h=pad, pad_w=pad, groups=groups, transposed=1, flip=0, math=0)
    k, launches = ctypes.c_int(-1), ctypes.c_int(-1)
    assert _native.lib().pasta_conv2d_plan(ctypes.byref(d), flags, None, None, None, ctypes.byref(launches), ctypes.byref(k)) == 0
    return k.value, launches.value


@pytest.mark.parametrize('n,ci,h,co,groups', [
    (2, 32, 64, 64, 1),          # two regular tiles per image row, one row tile, one column tile
    (9, 16, 64, 40, 1),          # nine images: the second row / column tile holds one image; 40 of 64 output rows
    (3, 48, 64, 96, 1),          # two output tiles, the second half empty; a channel count that is not a multiple of 32
    (2, 64, 64, 128, 2),         # groups
    (1, 24, 128, 72, 1),         # 128 x 128: five blocks of column rows; ragged channels on both sides
    (3, 32, 80, 64, 1),          # 80 x 80: no 32-column tiles -- the 16 x 16 tile instance (a 32-pixel MFMA block = two tile rows)
    (17, 48, 16, 72, 1),         # 16 x 16 planes, seventeen images: the edge tiles of the 16 x 16 instance hold sixteen images each
    (5, 64, 32, 96, 1),          # 32 x 32 planes
])
def test_one_pass_kernel_against_fp64(n, ci, h, co, groups):
    from torch_utils.ops import conv2d_gradfix as cg
    assert _plan(n, ci, h, co, groups) == (13, 1)
    g = torch.Generator().manual_seed(n * 1000 + ci + co)
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    w = (torch.randn([ci, co // groups, 3, 3], generator=g) * 0.1).cuda()
    y = cg._launch_conv(x, w, cg._Cfg((True, 2, 0, 0, 0, 0, groups)))
    ref = torch.nn.functional.conv_transpose2d(x.double(), w.double(), stride=2, groups=groups)
    assert y.shape == ref.shape == (n, co, 2 * h + 1, 2 * h + 1)
    assert _rel(y, ref) < 2e-6
    # the remainder row, the remainder column and the corner on their own (1 % of the outputs: a whole-tensor maximum would hide them)
    assert _rel(y[:, :, -1], ref[:, :, -1]) < 2e-6 and _rel(y[:, :, :, -1], ref[:, :, :, -1]) < 2e-6 and _rel(y[:, :, -1, -1], ref[:, :, -1, -1]) < 4e-6


def test_input_scale_rides_in_the_staging_of_every_tile_kind():
    """The modulated layers of the training step (x * styles inside the kernel): regular, row-edge and column-edge tiles."""
    from torch_utils.ops import conv2d_gradfix as cg
    assert _plan(5, 64, 64, 64, flags=1) == (13, 1)
    g = torch.Generator().manual_seed(7)
    x = torch.randn([5, 64, 64, 64], generator=g).cuda()
    w = (torch.randn([64, 64, 3, 3], generator=g) * 0.1).cuda()
    s = (torch.rand([5, 64], generator=g) * 3 + 0.25).cuda()
    y = cg._launch_conv(x, w, cg._Cfg((True, 2, 0, 0, 0, 0, 1)), iscale=s)
    ref = torch.nn.functional.conv_transpose2d(x.double() * s.double()[:, :, None, None], w.double(), stride=2)
    assert _rel(y, ref) < 2e-6 and _rel(y[:, :, -1], ref[:, :, -1]) < 2e-6 and _rel(y[:, :, :, -1], ref[:, :, :, -1]) < 2e-6


def test_what_the_kernel_leaves_to_the_other_paths():
    assert _plan(2, 32, 64, 64, pad=1)[0] != 13          # pad 1: the parity-pair mode or the per-class launches
    assert _plan(16, 512, 32, 256) == (13, 1) and _plan(48, 512, 16, 512) == (13, 1)      # 32 x 32 and 16 x 16 planes: the 8 x 32 and the 16 x 16 tile instance
    assert _plan(2, 32, 8, 64)[0] != 13                  # planes below 16 x 16
    assert _plan(2, 8, 64, 64)[0] != 13                  # fewer than sixteen input channels
    assert _plan(2, 32, 64, 64, flags=2)[0] != 13        # an output scale


def test_operator_api_first_and_second_derivatives():
    """conv_transpose2d through conv2d_gradfix (the reference's operator, conv2d_gradfix.py:34-37) on a shape the kernel takes: forward, both
    first derivatives and a second derivative against fp64 autograd.  The input gradient is a stride-2 convolution, the weight gradient the
    stride-2 weight-gradient kernel: only the forward pass is this kernel's."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(11)
    x = torch.randn([2, 32, 64, 64], generator=g).cuda().requires_grad_(True)
    w = (torch.randn([32, 48, 3, 3], generator=g) * 0.1).cuda().requires_grad_(True)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    y = cg.conv_transpose2d(x, w, stride=2)
    yd = torch.nn.functional.conv_transpose2d(xd, wd, stride=2)
    assert _rel(y, yd) < 2e-6
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(12)).cuda()
    gx, gw = torch.autograd.grad(y, [x, w], dy, create_graph=True)
    gxd, gwd = torch.autograd.grad(yd, [xd, wd], dy.double(), create_graph=True)
    assert _rel(gx, gxd) < 2e-6 and _rel(gw, gwd) < 2e-6
    (ggw,) = torch.autograd.grad((gx * gx).sum(), [w])
    (ggwd,) = torch.autograd.grad((gxd * gxd).sum(), [wd])
    assert _rel(ggw, ggwd) < 5e-6


def test_input_gradient_of_a_stride_two_convolution_is_this_kernel():
    """dx of conv2d(x[257 x 257], w, stride 2) = conv_transpose2d(dy[128 x 128], w) onto 257 x 257: the discriminator's down path backwards."""
    from torch_utils.ops import conv2d_gradfix as cg
    assert _plan(2, 128, 128, 64) == (13, 1)
    g = torch.Generator().manual_seed(3)
    x = torch.randn([2, 64, 257, 257], generator=g).cuda().requires_grad_(True)
    w = (torch.randn([128, 64, 3, 3], generator=g) * 0.1).cuda()
    y = cg.conv2d(x, w, stride=2)
    dy = torch.randn(y.shape, generator=g).cuda()
    (gx,) = torch.autograd.grad(y, [x], dy)
    xd = x.detach().double().requires_grad_(True)
    (gxd,) = torch.autograd.grad(torch.nn.functional.conv2d(xd, w.double(), stride=2), [xd], dy.double())
    assert _rel(gx, gxd) < 2e-6 and _rel(gx[:, :, -1], gxd[:, :, -1]) < 2e-6 and _rel(gx[:, :, :, -1], gxd[:, :, :, -1]) < 2e-6


def test_weights_by_lds_dma_change_no_bit_of_the_tile_kernel():
    """PASTA_ROWS2D_GLDS=1 (opt-in: profiles/r5_ab_rows2d_glds.txt): the eight-wave 3x3 stride-1 tile kernel with its weights by LDS-DMA and
    hand-counted waits -- the same products in the same order, so the same bits.  The switch is read once per process: two child processes."""
    import os, subprocess, sys
    code = ("import sys, hashlib; sys.path.insert(0, %r); import torch; from torch_utils.ops import conv2d_gradfix as cg\n"
            "g = torch.Generator().manual_seed(5)\n"
            "x = torch.randn([3, 96, 64, 64], generator=g).cuda(); w = (torch.randn([256, 96, 3, 3], generator=g) * 0.1).cuda()\n"
            "y = cg._launch_conv(x, w, cg._Cfg((False, 1, 1, 1, 0, 0, 1)))\n"
            "print('HASH', hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())\n") % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pasta-gan_amd')
    out = []
    for v in ('0', '1'):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, PASTA_ROWS2D_GLDS=v), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith('HASH')][0])
    assert out[0] == out[1]
