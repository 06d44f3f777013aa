"""HIP op layer (through the C ABI) against the reference's golden vectors and the CPU oracle."""

import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu

# Stated tolerance of the parity bar (BASELINE.json north_star): 1e-3 relative, fp32.  The op-level
# checks below are held far tighter; only fp16 storage cases use the looser figure.
TOL = 1e-5
TOL_FP16 = 2e-3


def dev(a, grad=False, dtype=None):
    x = torch.from_numpy(np.asarray(a)).to('cuda')
    if dtype is not None:
        x = x.to(dtype)
    return x.requires_grad_(True) if grad else x


def _cases(fname):
    g = load_golden(fname)
    return g, json.loads(str(g['manifest']))


def test_native_library_is_loaded():
    from torch_utils.ops import _native
    lib = _native.lib()
    assert lib.pasta_abi_version() >= 11
    assert b'gfx950' in lib.pasta_build_info()


# ----------------------------------------------------------------------------- upfirdn2d

@pytest.mark.parametrize('idx', range(18))
def test_upfirdn2d_golden(idx):
    from torch_utils.ops import upfirdn2d
    g, cases = _cases('ops_upfirdn2d.npz')
    c = cases[idx]
    n = c['name']
    x = dev(g[n + '.x'], True)
    f = dev(g[n + '.f']) if c['f'] is not None else None
    y = upfirdn2d.upfirdn2d(x, f, **c['call'])
    assert rel_err(y, g[n + '.y']) < TOL, n
    dx, = torch.autograd.grad(y, x, dev(g[n + '.dy']))
    assert rel_err(dx, g[n + '.dx']) < TOL, n


LIVE_UPFIRDN = [   # shapes of SURVEY.md 2.2 / 8(a1), scaled down in N and C only
    dict(shape=[2, 8, 256, 256], kw=dict(padding=[2, 2, 2, 2])),
    dict(shape=[2, 8, 257, 257], kw=dict(padding=[1, 1, 1, 1], gain=4)),
    dict(shape=[2, 16, 129, 129], kw=dict(padding=[1, 1, 1, 1], gain=4)),
    dict(shape=[2, 3, 128, 128], kw=dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    dict(shape=[2, 8, 256, 256], kw=dict(down=2, padding=[1, 1, 1, 1])),
    dict(shape=[4, 32, 9, 9], kw=dict(padding=[1, 1, 1, 1], gain=4)),
    dict(shape=[4, 32, 8, 8], kw=dict(padding=[2, 2, 2, 2])),
    dict(shape=[2, 16, 33, 33], kw=dict(padding=[1, 1, 1, 1], gain=4)),
    dict(shape=[2, 16, 32, 32], kw=dict(down=2, padding=[1, 1, 1, 1])),
    dict(shape=[2, 3, 4, 4], kw=dict(up=2, padding=[2, 1, 2, 1], gain=4)),
]


@pytest.mark.parametrize('idx', range(len(LIVE_UPFIRDN)))
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_upfirdn2d_live_shapes(idx, dtype):
    from torch_utils.ops import upfirdn2d
    c = LIVE_UPFIRDN[idx]
    gen = torch.Generator().manual_seed(idx)
    xc = torch.randn(c['shape'], generator=gen)
    if dtype == torch.float16:
        xc = xc.half().float()
    f = R.setup_filter([1, 3, 3, 1])
    xr = xc.clone().requires_grad_(True)
    yr = R.upfirdn2d(xr, f, **c['kw'])
    dyc = torch.randn(yr.shape, generator=gen)
    if dtype == torch.float16:
        dyc = dyc.half().float()
    dxr, = torch.autograd.grad(yr, xr, dyc)
    x = xc.to('cuda', dtype).requires_grad_(True)
    y = upfirdn2d.upfirdn2d(x, f.cuda(), **c['kw'])
    assert y.dtype == dtype and y.shape == yr.shape
    dx, = torch.autograd.grad(y, x, dyc.to('cuda', dtype))
    tol = TOL if dtype == torch.float32 else TOL_FP16
    assert rel_err(y.float(), yr) < tol
    assert rel_err(dx.float(), dxr) < tol


NONSQUARE_UPFIRDN = [   # planes whose one dimension is k*tile + 1 and whose other ends inside a tile
    dict(shape=[2, 3, 19, 32], kw=dict(padding=[2, 2, 2, 2])),          # 20 x 33 on 32 x 32 tiles: remainder column
    dict(shape=[2, 3, 32, 19], kw=dict(padding=[2, 2, 2, 2])),          # 33 x 20: remainder row
    dict(shape=[2, 3, 45, 64], kw=dict(padding=[2, 2, 2, 2])),          # 46 x 65
    dict(shape=[1, 2, 130, 256], kw=dict(padding=[2, 2, 2, 2])),        # 131 x 257 on 64 x 16 tiles
    dict(shape=[1, 2, 256, 130], kw=dict(padding=[2, 2, 2, 2])),        # 257 x 131
    dict(shape=[2, 3, 33, 50], kw=dict(padding=[1, 1, 1, 1], gain=4)),
    dict(shape=[2, 3, 40, 24], kw=dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    dict(shape=[2, 3, 40, 66], kw=dict(down=2, padding=[1, 1, 1, 1])),
]


@pytest.mark.parametrize('idx', range(len(NONSQUARE_UPFIRDN)))
def test_upfirdn2d_nonsquare_planes_stay_inside_their_plane(idx):
    """The output is a view into a larger NaN-filled buffer: a store outside the plane (or past the tensor) shows."""
    from torch_utils.ops import _native
    from torch_utils.ops import upfirdn2d
    c = NONSQUARE_UPFIRDN[idx]
    gen = torch.Generator().manual_seed(100 + idx)
    xc = torch.randn(c['shape'], generator=gen)
    f = R.setup_filter([1, 3, 3, 1])
    yr = R.upfirdn2d(xc, f, **c['kw'])
    y = upfirdn2d.upfirdn2d(xc.cuda(), f.cuda(), **c['kw'])
    assert y.shape == yr.shape and rel_err(y, yr) < TOL
    # same launch through the C ABI into the middle of a canary buffer
    lib = _native.lib()
    n = yr.numel()
    buf = torch.full([3 * n], float('nan'), device='cuda')
    out = buf[n:2 * n].view(yr.shape)
    parts = torch.zeros([256], device='cuda')
    kw = c['kw']
    up, down = kw.get('up', 1), kw.get('down', 1)
    px0, px1, py0, py1 = kw['padding']
    x = xc.cuda()
    fc = f.cuda()
    import ctypes
    i32x4, i64x4, i32x2 = ctypes.c_int32 * 4, ctypes.c_int64 * 4, ctypes.c_int32 * 2
    rc = lib.pasta_upfirdn2d(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(fc.data_ptr()), ctypes.c_void_p(out.data_ptr()), 0,
                             i32x4(*x.shape), i64x4(*x.stride()), i32x2(*fc.shape), i32x4(*out.shape), i64x4(*out.stride()),
                             up, up, down, down, px0, px1, py0, py1, 0, ctypes.c_float(kw.get('gain', 1)),
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(parts.data_ptr()), None)
    assert rc == 0, lib.pasta_last_error()
    torch.cuda.synchronize()
    assert rel_err(out, yr) < TOL
    assert float(parts.max()) == float(out.abs().max()) and float(parts.min()) >= 0      # producer-side |max| of the output
    assert bool(torch.isnan(buf[:n]).all()) and bool(torch.isnan(buf[2 * n:]).all()), 'stores outside the output tensor'


def test_upfirdn2d_channels_last_and_fp64():
    from torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(7)
    xc = torch.randn([2, 6, 20, 18], generator=gen, dtype=torch.float64)
    f = R.setup_filter([1, 3, 3, 1])
    ref = R.upfirdn2d(xc, f, up=2, padding=[2, 1, 2, 1], gain=4)
    y = upfirdn2d.upfirdn2d(xc.cuda(), f.cuda(), up=2, padding=[2, 1, 2, 1], gain=4)
    assert y.dtype == torch.float64 and rel_err(y, ref) < 1e-12
    xcl = xc.float().cuda().contiguous(memory_format=torch.channels_last)
    ycl = upfirdn2d.upfirdn2d(xcl, f.cuda(), up=2, padding=[2, 1, 2, 1], gain=4)
    assert ycl.is_contiguous(memory_format=torch.channels_last)
    assert rel_err(ycl, ref) < TOL


def test_upfirdn2d_double_backward_and_wrappers():
    from torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(3)
    xc = torch.randn([1, 2, 12, 12], generator=gen)
    f = R.setup_filter([1, 3, 3, 1])
    for name in ['upsample2d', 'downsample2d', 'filter2d']:
        xr = xc.clone().requires_grad_(True)
        yr = getattr(R, name)(xr, f)
        v = torch.randn(yr.shape, generator=gen).requires_grad_(True)
        gr, = torch.autograd.grad(yr, xr, v, create_graph=True)
        ggr, = torch.autograd.grad(gr.square().sum(), v)
        x = xc.cuda().requires_grad_(True)
        y = getattr(upfirdn2d, name)(x, f.cuda())
        vg = v.detach().cuda().requires_grad_(True)
        gh, = torch.autograd.grad(y, x, vg, create_graph=True)
        ggh, = torch.autograd.grad(gh.square().sum(), vg)
        assert rel_err(y, yr) < TOL and rel_err(gh, gr) < TOL and rel_err(ggh, ggr) < TOL, name


def test_upfirdn2d_errors():
    from torch_utils.ops import upfirdn2d
    f = R.setup_filter([1, 3, 3, 1])
    with pytest.raises(RuntimeError):
        upfirdn2d.upfirdn2d(torch.zeros(1, 1, 4, 4), f)          # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        upfirdn2d.upfirdn2d(torch.zeros(1, 1, 2, 2).cuda(), f.cuda())   # output smaller than 1x1
    with pytest.raises(NotImplementedError):
        upfirdn2d.upfirdn2d(torch.zeros(1, 1, 8, 8).cuda(), f.cuda(), impl='ref')


# ----------------------------------------------------------------------------- bias_act

@pytest.mark.parametrize('idx', range(24))
def test_bias_act_golden(idx):
    from torch_utils.ops import bias_act
    g, cases = _cases('ops_bias_act.npz')
    c = cases[idx]
    n = c['name']
    x = dev(g[n + '.x'], True)
    b = dev(g[n + '.b'], True) if c['bias'] else None
    dy = dev(g[n + '.dy'], True)
    y = bias_act.bias_act(x, b, act=c['act'], **c['kw'])
    assert rel_err(y, g[n + '.y']) < TOL, n
    grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dy, create_graph=True)
    assert rel_err(grads[0], g[n + '.dx']) < TOL, n
    if b is not None:
        assert rel_err(grads[1], g[n + '.db']) < TOL, n
    gg = torch.autograd.grad(grads[0], [dy, x], dev(g[n + '.ddx']), allow_unused=True)
    assert rel_err(gg[0], g[n + '.g_dy']) < TOL, n
    g_x = gg[1] if gg[1] is not None else torch.zeros_like(x)
    assert rel_err(g_x, g[n + '.g_x']) < 2e-5 or float(np.abs(g[n + '.g_x']).max()) == 0.0, n


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.float64])
@pytest.mark.parametrize('shape', [[4, 64, 64, 64], [3, 7, 33, 31], [16, 512]])
def test_bias_act_large(shape, dtype):
    from torch_utils.ops import bias_act
    gen = torch.Generator().manual_seed(11)
    xc = torch.randn(shape, generator=gen).to(dtype)
    bc = torch.randn([shape[1]], generator=gen).to(dtype)
    xr = xc.double().requires_grad_(True)
    br = bc.double().requires_grad_(True)
    # fp16 storage rounds y onto the clamp value itself, which moves the gradient mask; clamp only in fp32/fp64.
    clamp = None if dtype == torch.float16 else 1.5
    yr = R.bias_act(xr, br, act='lrelu', gain=np.sqrt(2), clamp=clamp)
    dyc = torch.randn(shape, generator=gen).to(dtype)
    dxr, dbr = torch.autograd.grad(yr, [xr, br], dyc.double())
    x = xc.cuda().requires_grad_(True)
    b = bc.cuda().requires_grad_(True)
    y = bias_act.bias_act(x, b, act='lrelu', gain=np.sqrt(2), clamp=clamp)
    dx, db = torch.autograd.grad(y, [x, b], dyc.cuda())
    tol = {torch.float32: TOL, torch.float16: TOL_FP16, torch.float64: 1e-6}[dtype]
    assert y.dtype == dtype
    assert rel_err(y, yr) < tol and rel_err(dx, dxr) < tol and rel_err(db, dbr) < max(tol, 1e-4)


@pytest.mark.parametrize('act,gain,clamp', [('lrelu', np.sqrt(2), 1.0), ('relu', 1.0, None), ('linear', 0.5, None), ('linear', 1.0, 0.8)])
def test_bias_act_fused_grad_and_bias_grad(act, gain, clamp):
    """The one-pass (dx, db) kernel: several chunks per plane with a ragged last chunk, and its double backward
    (d/d dy of a functional of dx and db)."""
    from torch_utils.ops import bias_act
    shape = [2, 5, 96, 96]          # 9216 elements per plane = 2304 packs -> chunks of 1024, 1024, 256
    assert bias_act._grad_db_workspace(torch.empty(shape, device='cuda'), 1, bias_act.activation_funcs[act].cuda_idx) == 2 * 5 * 3 * 4
    gen = torch.Generator().manual_seed(5)
    xc, bc, dyc = torch.randn(shape, generator=gen), torch.randn([5], generator=gen), torch.randn(shape, generator=gen)
    wc, vc = torch.randn(shape, generator=gen), torch.randn([5], generator=gen)
    def run(bias_act_fn, x, b, dy, w, v):
        x.requires_grad_(True); b.requires_grad_(True); dy.requires_grad_(True)
        y = bias_act_fn(x, b, act=act, gain=gain, clamp=clamp)
        dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
        g_dy, = torch.autograd.grad((dx * w).sum() + (db * v).sum(), [dy])
        return y, dx, db, g_dy
    ref = run(R.bias_act, xc.double(), bc.double(), dyc.double(), wc.double(), vc.double())
    out = run(bias_act.bias_act, xc.cuda(), bc.cuda(), dyc.cuda(), wc.cuda(), vc.cuda())
    for name, a, r in zip(['y', 'dx', 'db', 'g_dy'], out, ref):
        assert rel_err(a, r) < (1e-4 if name == 'db' else TOL), name


def test_bias_act_errors():
    from torch_utils.ops import bias_act
    with pytest.raises(RuntimeError):
        bias_act.bias_act(torch.zeros(2, 3), torch.zeros(3))             # CPU tensor
    with pytest.raises(AssertionError):
        bias_act.bias_act(torch.zeros(2, 3).cuda(), torch.zeros(4).cuda())
    y = bias_act.bias_act(torch.zeros(0, 3).cuda(), torch.zeros(3).cuda(), act='relu')
    assert y.shape == (0, 3)


# ----------------------------------------------------------------------------- conv2d_resample / conv2d_gradfix

@pytest.mark.parametrize('idx', range(15))
def test_conv2d_resample_golden(idx):
    from torch_utils.ops import conv2d_resample, upfirdn2d
    g, cases = _cases('ops_conv2d_resample.npz')
    c = cases[idx]
    n = c['name']
    x, w = dev(g[n + '.x'], True), dev(g[n + '.w'], True)
    f = upfirdn2d.setup_filter(c['f']).cuda() if c.get('f') is not None else None
    y = conv2d_resample.conv2d_resample(x, w, f=f, **c['kw'])
    assert rel_err(y, g[n + '.y']) < TOL, n
    dx, dw = torch.autograd.grad(y, [x, w], dev(g[n + '.dy']))
    assert rel_err(dx, g[n + '.dx']) < TOL, n
    assert rel_err(dw, g[n + '.dw']) < TOL, n


LIVE_CONV = [   # (x shape, w shape, kwargs): the tile configurations and layer types of the G/D path
    ([2, 128, 32, 32], [128, 128, 3, 3], dict(padding=1)),                 # 128x128 tile
    ([2, 64, 64, 64], [64, 64, 3, 3], dict(padding=1)),                    # 64x256 tile
    ([2, 64, 32, 32], [3, 64, 1, 1], dict()),                              # ToRGB, 32x256 tile
    ([4, 512, 4, 4], [512, 512, 3, 3], dict(padding=1)),                   # 64x64 tile
    ([2, 513, 4, 4], [512, 513, 3, 3], dict(padding=1)),                   # D epilogue (mbstd channel)
    ([2, 3, 64, 64], [64, 3, 7, 7], dict(padding=3)),                      # spade encoder stem
    ([2, 6, 64, 64], [64, 6, 1, 1], dict()),                               # encoder stem
    ([2, 192, 32, 32], [128, 192, 1, 1], dict()),                          # merge conv
    ([2, 64, 64, 64], [128, 64, 3, 3], dict(down=2, padding=1)),           # strided conv
    ([2, 64, 64, 64], [128, 64, 1, 1], dict(down=2)),                      # D skip
    ([2, 128, 32, 32], [64, 128, 3, 3], dict(up=2, padding=1, flip_weight=False)),   # synthesis conv0
    ([1, 2 * 32, 16, 16], [2 * 48, 32, 3, 3], dict(padding=1, groups=2)),  # eval-mode grouped modconv
    ([1, 2 * 32, 16, 16], [2 * 48, 32, 3, 3], dict(up=2, padding=1, groups=2, flip_weight=False)),
]


@pytest.mark.parametrize('idx', range(len(LIVE_CONV)))
def test_conv2d_resample_live_shapes(idx):
    from torch_utils.ops import conv2d_resample
    xs, ws, kw = LIVE_CONV[idx]
    gen = torch.Generator().manual_seed(100 + idx)
    xc = torch.randn(xs, generator=gen)
    wc = torch.randn(ws, generator=gen) / np.sqrt(ws[1] * ws[2] * ws[3])
    f = R.setup_filter([1, 3, 3, 1])
    xr, wr = xc.clone().requires_grad_(True), wc.clone().requires_grad_(True)
    yr = R.conv2d_resample(xr, wr, f=f, **kw)
    dyc = torch.randn(yr.shape, generator=gen)
    dxr, dwr = torch.autograd.grad(yr, [xr, wr], dyc)
    x, w = xc.cuda().requires_grad_(True), wc.cuda().requires_grad_(True)
    y = conv2d_resample.conv2d_resample(x, w, f=f.cuda(), **kw)
    dx, dw = torch.autograd.grad(y, [x, w], dyc.cuda())
    assert rel_err(y, yr) < TOL and rel_err(dx, dxr) < TOL and rel_err(dw, dwr) < 5e-5


def test_conv2d_double_backward_r1_pattern():
    """grad of |d out / d x|^2 wrt the weight: the R1 penalty's path (loss_wo_flow_fullbody.py:246-254)."""
    from torch_utils.ops import conv2d_resample, conv2d_gradfix, bias_act
    gen = torch.Generator().manual_seed(5)
    xc = torch.randn([2, 8, 16, 16], generator=gen)
    w1c = torch.randn([16, 8, 3, 3], generator=gen) * 0.2
    w2c = torch.randn([16, 16, 3, 3], generator=gen) * 0.2
    f = R.setup_filter([1, 3, 3, 1])

    def run(x, w1, w2, conv, act, ctx):
        h = act(conv(x, w1, padding=1), act='lrelu')
        out = conv(h, w2, f=f.to(x.device), down=2, padding=1)
        with ctx():
            gx, = torch.autograd.grad(out.sum(), x, create_graph=True)
        pen = gx.square().sum()
        return (out, gx, pen) + torch.autograd.grad(pen, [w1, w2])

    import contextlib
    ref = run(xc.clone().requires_grad_(True), w1c.clone().requires_grad_(True), w2c.clone().requires_grad_(True),
              R.conv2d_resample, R.bias_act, contextlib.nullcontext)
    got = run(xc.cuda().requires_grad_(True), w1c.cuda().requires_grad_(True), w2c.cuda().requires_grad_(True),
              conv2d_resample.conv2d_resample, bias_act.bias_act, conv2d_gradfix.no_weight_gradients)
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 5e-5


def test_conv2d_gradfix_api():
    from torch_utils.ops import conv2d_gradfix
    gen = torch.Generator().manual_seed(9)
    x = torch.randn([2, 5, 11, 9], generator=gen)
    w = torch.randn([7, 5, 3, 3], generator=gen)
    b = torch.randn([7], generator=gen)
    ref = torch.nn.functional.conv2d(x, w, b, stride=2, padding=1)
    got = conv2d_gradfix.conv2d(x.cuda(), w.cuda(), b.cuda(), stride=2, padding=1)
    assert rel_err(got, ref) < TOL
    wt = torch.randn([5, 4, 3, 3], generator=gen)
    ref = torch.nn.functional.conv_transpose2d(x, wt, stride=2, padding=1, output_padding=1)
    got = conv2d_gradfix.conv_transpose2d(x.cuda(), wt.cuda(), stride=2, padding=1, output_padding=1)
    assert rel_err(got, ref) < TOL
    with pytest.raises(RuntimeError):
        conv2d_gradfix.conv2d(x.cuda(), torch.randn(7, 4, 3, 3).cuda())
    with pytest.raises(RuntimeError):
        conv2d_gradfix.conv2d(x, w)   # CPU tensors are refused


# ----------------------------------------------------------------------------- fma / plane kernels

def test_fma_golden():
    from torch_utils.ops import fma
    g = load_golden('ops_fma.npz')
    a, b, c = dev(g['a'], True), dev(g['b'], True), dev(g['c'], True)
    y = fma.fma(a, b, c)
    assert rel_err(y, g['y']) < TOL
    da, db, dc = torch.autograd.grad(y, [a, b, c], dev(g['dy']))
    assert rel_err(da, g['da']) < TOL and rel_err(db, g['db']) < TOL and rel_err(dc, g['dc']) < TOL


@pytest.mark.parametrize('hw', [(128, 128), (64, 64), (17, 13)])
def test_spade_norm(hw):
    from training import networks
    gen = torch.Generator().manual_seed(21)
    shape = [2, 5, hw[0], hw[1]]
    xc = torch.randn(shape, generator=gen) * 2 + 0.5
    gc = torch.randn(shape, generator=gen) * 0.3
    bc = torch.randn(shape, generator=gen) * 0.3
    dc = torch.randn(shape, generator=gen)
    xr, gr, br = [t.double().requires_grad_(True) for t in (xc, gc, bc)]
    yr = torch.nn.functional.instance_norm(xr, eps=1e-5) * (1 + gr) + br
    dr = torch.autograd.grad(yr, [xr, gr, br], dc.double())
    x, g_, b = [t.cuda().requires_grad_(True) for t in (xc, gc, bc)]
    y = networks.spade_modulate(x, g_, b)
    d = torch.autograd.grad(y, [x, g_, b], dc.cuda())
    assert rel_err(y, yr) < TOL
    for a, r in zip(d, dr):
        assert rel_err(a, r) < 2e-5


@pytest.mark.parametrize('hw', [(128, 128), (64, 64), (17, 13)])
@pytest.mark.parametrize('clamp', [None, 0.9])
def test_spade_norm_with_fused_activation(hw, clamp):
    """spade_modulate(relu_gain=, clamp=) == bias_act(relu, gain, clamp) applied to the plain result: the activation
    Spade_Conv2dLayer (networks.py:4346-4352) puts in front of its convolution, done in the normalisation pass."""
    from training import networks
    gen = torch.Generator().manual_seed(23)
    shape = [2, 3, hw[0], hw[1]]
    xc = torch.randn(shape, generator=gen) * 2 + 0.5
    gc = torch.randn(shape, generator=gen) * 0.3
    bc = torch.randn(shape, generator=gen) * 0.3
    dc = torch.randn(shape, generator=gen)
    gain = float(np.sqrt(2) * np.sqrt(0.5) * 1.1)
    xr, gr, br = [t.double().requires_grad_(True) for t in (xc, gc, bc)]
    yr = torch.relu(torch.nn.functional.instance_norm(xr, eps=1e-5) * (1 + gr) + br) * gain
    if clamp is not None:
        yr = yr.clamp(-clamp, clamp)
    dr = torch.autograd.grad(yr, [xr, gr, br], dc.double())
    x, g_, b = [t.cuda().requires_grad_(True) for t in (xc, gc, bc)]
    y = networks.spade_modulate(x, g_, b, relu_gain=gain, clamp=clamp)
    d = torch.autograd.grad(y, [x, g_, b], dc.cuda())
    assert rel_err(y, yr) < TOL
    for a, r in zip(d, dr):
        assert rel_err(a, r) < 2e-5
    # beta gradient alone (the kernel still has to run for the activation mask)
    db, = torch.autograd.grad(networks.spade_modulate(x.detach(), g_.detach(), b, relu_gain=gain, clamp=clamp), [b], dc.cuda())
    assert rel_err(db, dr[2]) < 2e-5


@pytest.mark.parametrize('case', [
    dict(x=[2, 16, 24, 24], w=[32, 16, 3, 3], kw=dict(padding=1), act='lrelu', gain=np.sqrt(2), clamp=0.8),
    dict(x=[2, 16, 24, 24], w=[32, 16, 3, 3], kw=dict(down=2, padding=1), act='lrelu', gain=1.0, clamp=256.0),
    dict(x=[2, 16, 24, 24], w=[24, 16, 1, 1], kw=dict(down=2), act='linear', gain=np.sqrt(0.5), clamp=None, bias=False),
    dict(x=[2, 3, 40, 40], w=[64, 3, 7, 7], kw=dict(padding=3), act='relu', gain=None, clamp=None),
    dict(x=[4, 512, 4, 4], w=[64, 512, 3, 3], kw=dict(padding=1), act='lrelu', gain=None, clamp=256.0),          # split-K reduce epilogue
    dict(x=[2, 16, 12, 12], w=[32, 16, 3, 3], kw=dict(up=2, padding=1, flip_weight=False), act='lrelu', gain=None, clamp=None),  # not fusable: falls back
])
def test_conv2d_resample_bias_act_fused(case):
    """Conv2dLayer's conv + bias_act with the epilogue fused into the convolution, incl. the R1-style double backward."""
    from torch_utils.ops import conv2d_resample
    gen = torch.Generator().manual_seed(77)
    xc = torch.randn(case['x'], generator=gen)
    wc = torch.randn(case['w'], generator=gen) / np.sqrt(case['w'][1] * case['w'][2] * case['w'][3])
    bc = torch.randn([case['w'][0]], generator=gen) * 0.3 if case.get('bias', True) else None
    f = R.setup_filter([1, 3, 3, 1])

    def run(x, w, b, f, fused):
        x = x.requires_grad_(True); w = w.requires_grad_(True)
        if b is not None:
            b = b.requires_grad_(True)
        if fused:
            y = conv2d_resample.conv2d_resample_bias_act(x, w, b, f=f, act=case['act'], gain=case['gain'], clamp=case['clamp'], **case['kw'])
        else:
            y = R.bias_act(R.conv2d_resample(x, w, f=f, **case['kw']), b, act=case['act'], gain=case['gain'], clamp=case['clamp'])
        gx, = torch.autograd.grad(y.square().sum(), x, create_graph=True)
        params = [w] + ([b] if b is not None else [])
        return [y, gx] + list(torch.autograd.grad(gx.square().sum() + y.sum(), params))

    ref = run(xc.clone(), wc.clone(), bc.clone() if bc is not None else None, f, False)
    got = run(xc.cuda(), wc.cuda(), bc.cuda() if bc is not None else None, f.cuda(), True)
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 5e-5


def test_nan_to_num_multi():
    """misc.nan_to_num_ == torch.nan_to_num per tensor (training_loop_wo_flow_fullbody.py:513-515): 230 tensors of ragged
    sizes (more than one launch's table), unaligned views, an empty tensor, sizes around the 16 K chunk."""
    from torch_utils import misc
    gen = torch.Generator().manual_seed(77)
    sizes = [0, 1, 3, 4, 5, 1023, 16383, 16384, 16385, 40000, 3 * 512 * 9, 512 * 512 * 9] + [int(s) for s in torch.randint(1, 5000, [218], generator=gen)]
    ts = []
    for k, n in enumerate(sizes):
        t = torch.randn([n + 1], generator=gen)
        if n:
            idx = torch.randint(0, n + 1, [max(1, n // 7)], generator=gen)
            t[idx[0::3]] = float('nan'); t[idx[1::3]] = float('inf'); t[idx[2::3]] = -float('inf')
        t = t.cuda()
        ts.append(t[1:] if k % 2 else t[:n])            # odd ones: 4-byte aligned only
    want = [torch.nan_to_num(t, nan=0, posinf=1e5, neginf=-1e5) for t in ts]
    misc.nan_to_num_(ts, nan=0, posinf=1e5, neginf=-1e5)
    for t, w in zip(ts, want):
        assert torch.equal(t, w)
    # defaults: nan -> 0, infinities -> the largest finite values
    t = torch.tensor([float('nan'), float('inf'), -float('inf'), 1.5], device='cuda')
    misc.nan_to_num_([t])
    assert t[0] == 0 and t[1] == torch.finfo(torch.float32).max and t[2] == -torch.finfo(torch.float32).max and t[3] == 1.5


@pytest.mark.parametrize('case', [
    dict(x=[2, 48, 32, 32], w=[40, 48, 1, 1], kw=dict(), act='lrelu', clamp=0.7),                  # 1x1, 128-pixel rows: split-bf16 base kernel
    dict(x=[2, 16, 24, 24], w=[24, 16, 3, 3], kw=dict(padding=1), act='linear', clamp=None),       # fp32 tile kernel
    dict(x=[4, 128, 32, 32], w=[128, 128, 3, 3], kw=dict(padding=1), act='linear', clamp=None),    # row-reuse kernel
    dict(x=[4, 512, 4, 4], w=[64, 512, 3, 3], kw=dict(padding=1), act='relu', clamp=None),         # split-K reduce epilogue
])
def test_conv_residual_epilogue(case):
    """conv2d_bias_act(..., residual=r) == bias_act(conv(x, w) + r, b): forward, and gradients of x, w, b and r."""
    from torch_utils.ops import conv2d_gradfix as cg
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(case['x'], generator=gen)
    w = torch.randn(case['w'], generator=gen) / np.sqrt(np.prod(case['w'][1:]))
    b = torch.randn([case['w'][0]], generator=gen) * 0.1
    ref_in = [t.double().requires_grad_(True) for t in (x, w, b)]
    y0 = torch.nn.functional.conv2d(ref_in[0], ref_in[1], **case['kw'])
    r = torch.randn(y0.shape, generator=gen)
    r64 = r.double().requires_grad_(True)
    z = y0 + r64 + ref_in[2].reshape(1, -1, 1, 1)
    gain = 1.3
    yr = {'linear': z, 'relu': torch.relu(z), 'lrelu': torch.nn.functional.leaky_relu(z, 0.2)}[case['act']] * gain
    if case['clamp'] is not None:
        yr = yr.clamp(-case['clamp'], case['clamp'])
    dy = torch.randn(yr.shape, generator=gen)
    gr = torch.autograd.grad(yr, ref_in + [r64], dy.double())
    xs = [t.cuda().requires_grad_(True) for t in (x, w, b, r)]
    y = cg.conv2d_bias_act(xs[0], xs[1], xs[2], act=case['act'], gain=gain, clamp=case['clamp'], residual=xs[3], **case['kw'])
    assert rel_err(y, yr) < TOL
    g = torch.autograd.grad(y, xs, dy.cuda())
    for a, ref in zip(g, gr):
        assert rel_err(a, ref) < 2e-5


@pytest.mark.parametrize('hw', [(128, 128), (64, 64), (17, 13)])
@pytest.mark.parametrize('relu_gain', [None, 1.1])
def test_spade_norm_with_gamma_beta_halves(hw, relu_gain):
    """spade_modulate(x, gb, None): gamma | beta as the channel halves of one tensor (the output of one convolution with the
    concatenated conv_gamma / conv_beta weights) == spade_modulate(x, gamma, beta); the gradient comes back as one tensor."""
    from training import networks
    gen = torch.Generator().manual_seed(29)
    n, c = 2, 3
    x = (torch.randn([n, c, *hw], generator=gen) * 2 + 0.5).cuda().requires_grad_(True)
    gb = (torch.randn([n, 2 * c, *hw], generator=gen) * 0.3).cuda().requires_grad_(True)
    dy = torch.randn([n, c, *hw], generator=gen).cuda()
    kw = dict(relu_gain=relu_gain, clamp=0.9) if relu_gain is not None else {}
    y1 = networks.spade_modulate(x, gb[:, :c], gb[:, c:], **kw)
    g1 = torch.autograd.grad(y1, [x, gb], dy)
    y2 = networks.spade_modulate(x, gb, None, **kw)
    g2 = torch.autograd.grad(y2, [x, gb], dy)
    assert torch.equal(y1, y2)
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    dx_only, = torch.autograd.grad(networks.spade_modulate(x, gb.detach(), None, **kw), [x], dy)
    assert torch.equal(dx_only, g2[0])


# ----------------------------------------------------------------------------- tensor_amax (operand scales of PASTA_MATH_F16X3)

@pytest.mark.parametrize('numel,offset', [(1, 0), (255, 0), (1 << 20, 0), ((1 << 22) + 3, 1), (16 * 64 * 129 * 129, 3)])
def test_tensor_amax_partial_maxima(numel, offset):
    """256 partial maxima whose maximum is the tensor's largest FINITE magnitude; any alignment (views into a larger tensor)."""
    from torch_utils.ops import _native
    lib = _native.lib()
    g = torch.Generator().manual_seed(numel % 9973)
    base = torch.randn([numel + offset], generator=g).cuda()
    x = base[offset:]
    x[numel // 2] = -77.5
    if numel > 4:
        x[1] = float('inf'); x[numel - 2] = float('nan'); x[3] = float('-inf')        # skipped by the scan
    parts = torch.full([256], -1.0, device='cuda')
    _native.check(lib.pasta_tensor_amax(_native.ptr(x), numel, 0, _native.ptr(parts), _native.stream()))
    assert float(parts.min()) >= 0
    finite = x[torch.isfinite(x)]
    assert float(parts.max()) == float(finite.abs().max()) == 77.5
    # cached front end (round 4: only on tensors no write can reach behind the version counter -- tests/test_f16x3_hardening_gpu.py)
    from torch_utils.ops import conv2d_gradfix as cg
    t = torch.randn([4, 8, 16, 16], generator=g).cuda().requires_grad_(True) * 1.0
    p1 = cg.tensor_amax(t)
    assert cg.tensor_amax(t) is p1                      # same version: no second scan
    with torch.no_grad():
        t.mul_(2)
    p2 = cg.tensor_amax(t)
    assert p2 is not p1 and float(p2.max()) == float(t.abs().max())


def test_producer_side_maxima_match_a_scan():
    """Every operator that writes activations leaves its output's largest magnitude behind (include/pasta_hip.h, "producer-side
    maxima"); conv2d_gradfix.tensor_amax then finds it on the tensor instead of scanning.  Here: each producer's row against
    the tensor's true |max|, and a convolution fed by each against fp64."""
    from torch_utils.ops import bias_act, upfirdn2d, fma, conv2d_gradfix as cg
    from training import networks
    g = torch.Generator().manual_seed(21)
    x = (torch.randn([4, 32, 64, 64], generator=g) * 3).cuda()          # 16384 pixels: the matrix-core tiles (and the three-product arithmetic) run
    b = torch.randn([32], generator=g).cuda()
    f = R.setup_filter([1, 3, 3, 1]).cuda()
    s = torch.randn([4, 32], generator=g).cuda()
    outs = {
        'bias_act': bias_act.bias_act(x, b, act='lrelu', gain=2 ** 0.5, clamp=256),
        'upfirdn2d': upfirdn2d.upfirdn2d(x, f, padding=[2, 2, 2, 2]),
        'upfirdn2d_up': upfirdn2d.upsample2d(x, f),
        'scale_planes': fma.scale_planes(x, s),
        'spade': networks.spade_modulate(x, x * 0.1, x * 0.2),
        'mod_bias_act': networks.mod_bias_act(x, s.abs(), None, None, b, act='lrelu', clamp=256),
        'conv_epilogue': cg.conv2d_bias_act(x, torch.randn([48, 32, 3, 3], generator=g).cuda() / 17, b.new_zeros(48), padding=1, act='lrelu'),
    }
    w = (torch.randn([64, 32, 3, 3], generator=g) / 17).cuda()
    for name, t in outs.items():
        hit = getattr(t, '_pasta_amax', None)
        if name in ('scale_planes', 'mod_bias_act'):       # thousands of four-instruction waves: a commit per wave costs what the scan costs; the consumer scans
            assert hit is None, name
            assert float(cg.tensor_amax(t).max()) == float(t.abs().max()), name
        else:
            assert hit is not None and hit[0] == t._version and hit[1] == t.data_ptr(), name
            assert float(hit[2].max()) == float(t.abs().max()), name
            assert cg.tensor_amax(t) is hit[2], name                   # the convolution takes the producer's row
        if t.shape[1] == 32:
            y = cg.conv2d(t, w, padding=1)
            ref = torch.nn.functional.conv2d(t.double(), w.double(), padding=1)
            assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6, name
    # backward producers: the fused bias_act gradient, the SPADE and mod_bias_act backward kernels
    xg = x.clone().requires_grad_(True)
    y = cg.conv2d_bias_act(xg, w, b.new_zeros(64), padding=1, act='lrelu')
    seen = []
    orig = cg.tensor_amax
    def spy(t):
        seen.append(getattr(t, '_pasta_amax', None) is not None and t._pasta_amax[0] == t._version)
        return orig(t)
    cg.tensor_amax = spy
    try:
        y.square().mean().backward()
    finally:
        cg.tensor_amax = orig
    assert seen == [True], seen      # the input-gradient launch: dz arrives with its producer's maxima; the weight is never scanned (its packing kernel scales it per row)
    # the SPADE backward with gamma | beta as halves of one tensor: dgamma | dbeta (the dy of ONE convolution's backward) carries its own row
    xs = x.clone().requires_grad_(True)
    gb = torch.cat([x * 0.1, x * 0.2], dim=1).requires_grad_(True)
    out = networks.spade_modulate(xs, gb, None, relu_gain=2 ** 0.5, clamp=256)
    dxs, dgb = torch.autograd.grad(out, [xs, gb], torch.randn(out.shape, generator=g).cuda())
    for name, t in (('spade dx', dxs), ('spade dgamma|dbeta', dgb)):
        hit = getattr(t, '_pasta_amax', None)
        assert hit is not None and hit[0] == t._version and hit[1] == t.data_ptr(), name
        assert float(hit[2].max()) == float(t.abs().max()), name


# ----------------------------------------------------------------------------- garment features of the SPADE stage

@pytest.mark.parametrize('hw', [(128, 128), (64, 64), (24, 40)])
def test_garment_feature_fill_matches_the_reference_expression(hw):
    """``cat([fill(feat_u), fill(feat_l)], 1)``, ``fill(x) = x (1 - hole) + (sum_hw(x valid) / count) hole`` (networks.py:5777-5800, 5836)
    by ``pasta_masked_mean_fill`` against the same expression in float64 torch operations: values, both feature gradients, the row of maxima."""
    from training import networks
    h, w = hw
    g = torch.Generator().manual_seed(h + w)
    n, c = 3, 8
    def garment():
        feat = torch.randn([n, c, h, w], generator=g)
        valid = (torch.rand([n, 1, h, w], generator=g) > 0.6).float()
        hole = ((torch.rand([n, 1, h, w], generator=g) > 0.7).float() * (1 - valid))
        valid[1] = 0                                        # a sample without a covered pixel: the divisor is the plane size
        count = valid.sum(dim=(2, 3), keepdim=True)
        enough = (count > 10).float()
        count = count * enough + float(h * w) * (1 - enough)
        return feat, valid, hole, count
    pu, pl = garment(), garment()
    def ref(parts):
        feat, valid, hole, count = (t.double() for t in parts)
        feat.requires_grad_(True)
        total = (feat * valid).sum(dim=(2, 3), keepdim=True)
        return feat, feat * (1 - hole) + (total / count) * hole
    fu, ou = ref(pu); fl, ol = ref(pl)
    want = torch.cat([ou, ol], dim=1)
    dy = torch.randn(want.shape, generator=g, dtype=torch.float64)
    gu, gl = torch.autograd.grad(want, [fu, fl], dy)
    cu = [t.cuda() for t in pu]; cl = [t.cuda() for t in pl]
    cu[0].requires_grad_(True); cl[0].requires_grad_(True)
    out = networks._GarmentFeat.apply(*cu, *cl)
    assert out.shape == want.shape and rel_err(out, want) < TOL
    hit = getattr(out, '_pasta_amax', None)
    if hit is not None:
        assert float(hit[2].max()) == float(out.abs().max())
    du, dl = torch.autograd.grad(out, [cu[0], cl[0]], dy.float().cuda())
    assert rel_err(du, gu) < TOL and rel_err(dl, gl) < TOL
    # a gradient of the gradient (create_graph): the differentiable fallback of the backward
    cu[0].grad = None
    out2 = networks._GarmentFeat.apply(*cu, *cl)
    v = dy.float().cuda().requires_grad_(True)
    d1, = torch.autograd.grad(out2, [cu[0]], v, create_graph=True)
    dd, = torch.autograd.grad(d1.square().sum(), [v])
    assert rel_err(d1, gu) < TOL and torch.isfinite(dd).all()
