"""Accuracy of the matrix-core arithmetics of the convolution against an fp64 reference: the fp32-equivalent modes --
PASTA_MATH_F16X3 (the default since round 3: three fp16 products of power-of-two-scaled hi / lo pieces) and
PASTA_MATH_BF16X6 (six bf16 products) -- must be as accurate as fp32 FMA chains, not merely inside the 1e-3 parity bar."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _errs(mode, x, w, ref64):
    from torch_utils.ops import conv2d_gradfix as cg
    old = cg.conv_math
    cg.conv_math = mode
    try:
        y = cg.conv2d(x, w, padding=1)
    finally:
        cg.conv_math = old
    d = (y.double().cpu() - ref64).abs()
    return float(d.max() / ref64.abs().max()), float(d.pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt())


@pytest.mark.parametrize('scale', [1.0, 1e-3, 300.0])
def test_split_bf16_matches_fp32_accuracy(scale):
    g = torch.Generator().manual_seed(0)
    x = torch.randn([8, 256, 40, 40], generator=g) * scale       # 12800 pixels x 128 channels -> the 128x128 tile
    w = torch.randn([128, 256, 3, 3], generator=g) / 48
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    e32 = _errs('f32', x.cuda(), w.cuda(), ref64)
    e16 = _errs('bf16x6', x.cuda(), w.cuda(), ref64)
    eh3 = _errs('f16x3', x.cuda(), w.cuda(), ref64)
    cpu = torch.nn.functional.conv2d(x, w, padding=1).double()
    ecpu = float((cpu - ref64).abs().max() / ref64.abs().max())
    print(f'scale {scale}: max-rel err fp32-MFMA {e32[0]:.3e}, split-bf16 {e16[0]:.3e}, fp16 x 3 {eh3[0]:.3e}, torch CPU fp32 {ecpu:.3e}; rms {e32[1]:.3e} vs {e16[1]:.3e} vs {eh3[1]:.3e}')
    assert e32[0] < 5e-6 and e16[0] < 5e-6 and eh3[0] < 5e-6         # K = 2304 products per output
    assert e16[1] < 3 * e32[1] + 1e-9          # same error class as fp32 arithmetic
    # the three-product arithmetic is held tighter: rms <= 1e-6 against fp64 and no worse than an fp32 FMA
    # chain (measured 5.3e-7 against 8.5e-7: three accumulations per K step instead of eight)
    assert eh3[1] < 1e-6 and eh3[1] < 1.2 * e32[1] + 1e-9


def test_math_modes_reach_their_kernels_and_agree():
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(1)
    x = torch.randn([16, 128, 32, 32], generator=g).cuda()       # 16384 pixels, 128 output channels -> 128x128 tile
    w = (torch.randn([128, 128, 3, 3], generator=g) / 34).cuda()
    out = {}
    for mode in ['f32', 'bf16x6', 'f16x3', 'default']:
        cg.conv_math = mode
        out[mode] = cg.conv2d(x, w, padding=1)
    cg.conv_math = 'default'
    assert torch.equal(out['default'], out['f16x3'])              # round 3: the default is the three-product fp16 arithmetic
    assert not torch.equal(out['f32'], out['bf16x6']) and not torch.equal(out['f16x3'], out['bf16x6'])      # different arithmetics really ran
    assert float((out['f32'] - out['bf16x6']).abs().max() / out['f32'].abs().max()) < 2e-6
    assert float((out['f32'] - out['f16x3']).abs().max() / out['f32'].abs().max()) < 2e-6


def _grads(mode, x, w, dy):
    from torch_utils.ops import conv2d_gradfix as cg
    old = cg.conv_math
    cg.conv_math = mode
    try:
        x = x.clone().requires_grad_(True)
        w = w.clone().requires_grad_(True)
        gx, gw = torch.autograd.grad(cg.conv2d(x, w, padding=1), [x, w], dy)
    finally:
        cg.conv_math = old
    return gx.double().cpu(), gw.double().cpu()


def test_split_bf16_gradients_match_fp32_accuracy():
    """Input gradient (split-bf16 forward kernel on the flipped weights) and weight gradient
    (conv_wgrad3x3_bf16x6_kernel, K = 8192 pixels per output) against fp64 autograd."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn([8, 128, 32, 32], generator=g)
    w = torch.randn([128, 128, 3, 3], generator=g) / 34
    dy = torch.randn([8, 128, 32, 32], generator=g)
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    rx, rw = torch.autograd.grad(torch.nn.functional.conv2d(x64, w64, padding=1), [x64, w64], dy.double())
    res = {}
    for mode in ['f32', 'bf16x6', 'f16x3']:
        gx, gw = _grads(mode, x.cuda(), w.cuda(), dy.cuda())
        res[mode] = (float((gx - rx).abs().max() / rx.abs().max()), float((gw - rw).abs().max() / rw.abs().max()),
                     float((gx - rx).pow(2).mean().sqrt() / rx.pow(2).mean().sqrt()), float((gw - rw).pow(2).mean().sqrt() / rw.pow(2).mean().sqrt()))
    print('max-rel (dx, dw), rms (dx, dw):', res)
    for mode in res:
        assert res[mode][0] < 5e-6 and res[mode][1] < 1e-5
    assert res['bf16x6'][2] < 3 * res['f32'][2] + 1e-9 and res['bf16x6'][3] < 3 * res['f32'][3] + 1e-9
    assert res['f16x3'][2] < 1e-6 and res['f16x3'][3] < 1e-6
    assert res['f16x3'][2] < 1.2 * res['f32'][2] + 1e-9 and res['f16x3'][3] < 1.2 * res['f32'][3] + 1e-9


@pytest.mark.parametrize('n,cin,cout,hw', [(9, 32, 48, 32), (2, 16, 136, 64), (1, 24, 40, 128), (1, 16, 64, 256), (3, 48, 64, 64), (2, 32, 130, 128),
                                           (4, 64, 128, 32), (2, 20, 200, 96)])
def test_row_reuse_kernel_every_tiling(n, cin, cout, hw):
    """conv_fwd_rows_bf16x6_kernel / conv_fwd_rows2d_bf16x6_kernel on every (tile, row-segment) combination: 128- and 256-pixel tiles made of 1, 2, 4
    or 8 row segments, image borders on all sides, channel tails; forward (ascending taps) and input gradient
    (descending taps) against torch's CPU convolution.  PASTA_ROWS2D=0 (tests/test_conv_rows2d_gpu.py runs it in a child) sends the same shapes to the row kernel."""
    import ctypes
    from torch_utils.ops import conv2d_gradfix as cg
    from torch_utils import custom_ops
    g = torch.Generator().manual_seed(hw + cout)
    x = torch.randn([n, cin, hw, hw], generator=g)
    w = torch.randn([cout, cin, 3, 3], generator=g) / (3 * cin ** 0.5)
    dy = torch.randn([n, cout, hw, hw], generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, w, padding=1)
    gxr, = torch.autograd.grad(yr, xr, dy)
    xc = x.cuda().requires_grad_(True)
    y = cg.conv2d(xc, w.cuda(), padding=1)
    gx, = torch.autograd.grad(y, xc, dy.cuda())
    assert float((y.detach().cpu() - yr.detach()).abs().max() / yr.detach().abs().max()) < 1e-5
    assert float((gx.cpu() - gxr).abs().max() / gxr.abs().max()) < 1e-5
    # which kernel ran
    desc = custom_ops.ConvDesc(N=n, C_in=cin, H=hw, W=hw, C_out=cout, OH=hw, OW=hw, kh=3, kw=3, stride=1, pad_h=1, pad_w=1, groups=1,
                               transposed=0, flip=0, math=0)
    kernel = ctypes.c_int()
    custom_ops.get_plugin().pasta_conv2d_plan(ctypes.byref(desc), 0, None, None, None, None, ctypes.byref(kernel))
    # more than 64 output channels (128 x 128 tile): the 2-D tiles of four rows (4); else (64 x 256 tile) those of eight rows (6).
    # (96-pixel rows are no whole number of the row kernel's power-of-two segments; the 2-D tiles take them as three 32-column blocks.)
    import os
    if os.environ.get('PASTA_ROWS2D') == '0':       # the row kernel; 96-pixel rows fall back to the base kernel
        assert kernel.value == (1 if hw == 96 else 2)
    else:                                            # 128-row tiles: eight waves on 128 x 256 (7); 64-row tiles: eight-row tiles (6)
        assert kernel.value == (7 if cout > 64 else 6)


@pytest.mark.parametrize('transposed,n,cin,cout,hw,pad', [(False, 2, 40, 72, 65, 0), (False, 2, 64, 64, 64, 1), (False, 1, 24, 130, 129, 0),
                                                        (True, 2, 72, 40, 32, 0), (True, 1, 130, 24, 64, 0), (False, 2, 32, 64, 33, 0)])
def test_stride2_weight_gradient_split_bf16(transposed, n, cin, cout, hw, pad):
    """conv_wgrad3x3s2_bf16x6_kernel: stride-2 3x3 weight gradients of conv2d (pad 0 on odd planes as in the
    discriminator's down blocks, pad 1 on even planes as in the encoders) and of conv_transpose2d (operand roles
    swapped), channel tails, against fp64 autograd; the last case (16-pixel rows... 33 -> 16) exercises K slicing."""
    import ctypes
    from torch_utils.ops import conv2d_gradfix as cg
    from torch_utils import custom_ops
    g = torch.Generator().manual_seed(hw * 7 + cout)
    x = torch.randn([n, cin, hw, hw], generator=g)
    if transposed:
        w = torch.randn([cin, cout, 3, 3], generator=g) / (3 * cin ** 0.5)
        ref_fn = lambda a, b: torch.nn.functional.conv_transpose2d(a, b, stride=2, padding=pad)
        our_fn = lambda a, b: cg.conv_transpose2d(a, b, stride=2, padding=pad)
    else:
        w = torch.randn([cout, cin, 3, 3], generator=g) / (3 * cin ** 0.5)
        ref_fn = lambda a, b: torch.nn.functional.conv2d(a, b, stride=2, padding=pad)
        our_fn = lambda a, b: cg.conv2d(a, b, stride=2, padding=pad)
    w64 = w.double().requires_grad_(True)
    y64 = ref_fn(x.double(), w64)
    dy = torch.randn(list(y64.shape), generator=g)
    rw, = torch.autograd.grad(y64, w64, dy.double())
    res = {}
    for mode in ['f32', 'bf16x6', 'f16x3']:
        old = cg.conv_math
        cg.conv_math = mode
        try:
            wc = w.cuda().requires_grad_(True)
            gw, = torch.autograd.grad(our_fn(x.cuda(), wc), wc, dy.cuda())
        finally:
            cg.conv_math = old
        res[mode] = float((gw.double().cpu() - rw).abs().max() / rw.abs().max())
    assert res['f32'] < 1e-5 and res['bf16x6'] < 1e-5 and res['f16x3'] < 1e-5, res
    oh = y64.shape[2]
    desc = custom_ops.ConvDesc(N=n, C_in=cin, H=hw, W=hw, C_out=cout, OH=oh, OW=oh, kh=3, kw=3, stride=2, pad_h=pad, pad_w=pad, groups=1,
                               transposed=int(transposed), flip=0, math=0, wscale=1.0)
    kernel = ctypes.c_int()
    custom_ops.get_plugin().pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(kernel))
    assert kernel.value == 3


@pytest.mark.parametrize('n,cin,cout,transposed', [(5, 72, 40, False), (16, 512, 512, False), (3, 48, 136, True)])
def test_sixteen_pixel_rows_run_the_split_weight_gradient_half_filled(n, cin, cout, transposed):
    """Round 5 (conv_igemm.hip, wgrad_wide16): 3x3 stride-1 weight gradients over 16 x 16 planes -- the 512-channel layers of both networks --
    on conv_wgrad3x3_bf16x6_kernel with a row taken as a 32-pixel chunk whose second half is masked to zero (they ran on the fp32-MFMA kernel):
    every arithmetic against fp64 autograd, channel tails, the transposed operator, the plan."""
    import ctypes
    from torch_utils.ops import conv2d_gradfix as cg
    from torch_utils import custom_ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn([n, cin, 16, 16], generator=g)
    w = torch.randn([cin, cout, 3, 3] if transposed else [cout, cin, 3, 3], generator=g) / (9 * cin) ** 0.5
    dy = torch.randn([n, cout, 16, 16], generator=g)
    ref_op = torch.nn.functional.conv_transpose2d if transposed else torch.nn.functional.conv2d
    w64 = w.double().requires_grad_(True)
    rw, = torch.autograd.grad(ref_op(x.double(), w64, padding=1), w64, dy.double())
    res = {}
    for mode in ['f32', 'bf16x6', 'f16x3']:
        old = cg.conv_math
        cg.conv_math = mode
        try:
            wc = w.cuda().requires_grad_(True)
            op = cg.conv_transpose2d if transposed else cg.conv2d
            gw, = torch.autograd.grad(op(x.cuda(), wc, padding=1), wc, dy.cuda())
        finally:
            cg.conv_math = old
        res[mode] = float((gw.double().cpu() - rw).abs().max() / rw.abs().max())
    assert res['f32'] < 1e-5 and res['bf16x6'] < 1e-5 and res['f16x3'] < 1e-5, res
    desc = custom_ops.ConvDesc(N=n, C_in=cin, H=16, W=16, C_out=cout, OH=16, OW=16, kh=3, kw=3, stride=1, pad_h=1, pad_w=1, groups=1,
                               transposed=int(transposed), flip=0, math=0, wscale=1.0)
    kernel = ctypes.c_int()
    custom_ops.get_plugin().pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(kernel))
    assert kernel.value == 2
    desc.H = desc.W = desc.OH = desc.OW = 8          # a quarter filled would not pay: 8-pixel rows stay on the fp32-MFMA kernel
    custom_ops.get_plugin().pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(kernel))
    assert kernel.value == 0


@pytest.mark.parametrize('n,cin,cout,hw', [(2, 192, 128, 64), (2, 64, 40, 32), (1, 24, 200, 128), (3, 130, 130, 16), (16, 512, 512, 8), (16, 512, 512, 4)])
def test_pointwise_weight_gradient_split_bf16(n, cin, cout, hw):
    """conv_wgrad1x1_bf16x6_kernel: 128x128 and 64x64 channel tiles, channel tails, small planes (K slicing over few
    chunks), against fp64 autograd."""
    import ctypes
    from torch_utils.ops import conv2d_gradfix as cg
    from torch_utils import custom_ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn([n, cin, hw, hw], generator=g)
    w = torch.randn([cout, cin, 1, 1], generator=g) / cin ** 0.5
    dy = torch.randn([n, cout, hw, hw], generator=g)
    w64 = w.double().requires_grad_(True)
    rw, = torch.autograd.grad(torch.nn.functional.conv2d(x.double(), w64), w64, dy.double())
    res = {}
    for mode in ['f32', 'bf16x6', 'f16x3']:
        old = cg.conv_math
        cg.conv_math = mode
        try:
            wc = w.cuda().requires_grad_(True)
            gw, = torch.autograd.grad(cg.conv2d(x.cuda(), wc), wc, dy.cuda())
        finally:
            cg.conv_math = old
        res[mode] = float((gw.double().cpu() - rw).abs().max() / rw.abs().max())
    assert res['f32'] < 1e-5 and res['bf16x6'] < 1e-5 and res['f16x3'] < 1e-5, res
    desc = custom_ops.ConvDesc(N=n, C_in=cin, H=hw, W=hw, C_out=cout, OH=hw, OW=hw, kh=1, kw=1, stride=1, pad_h=0, pad_w=0, groups=1,
                               transposed=0, flip=0, math=0, wscale=1.0)
    kernel = ctypes.c_int()
    custom_ops.get_plugin().pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(kernel))
    assert kernel.value == (4 if hw * hw % 32 == 0 else 0)       # 4 x 4 planes stay on the fp32 kernel


# ---------------------------------------------------------------------------------------------------------------------
# Opt-in reduced-product modes (PASTA_MATH_BF16X3 = the counterpart of the reference's allow_tf32, PASTA_MATH_BF16 = plain
# bf16 operands).  Same kernels with fewer pieces per operand; the bounds are those of the arithmetic: dropped terms of
# relative size 2^-16 (three of them) resp. operand rounding of 2^-9, over K random-sign products.

REDUCED = {'bf16x3': (2e-5, 1.5e-7), 'bf16': (6e-3, 2e-4)}    # mode: (upper bound, lower bound) on the rms relative error (the six-product mode measures 2-5e-8)


def _conv_case(kind):
    g = torch.Generator().manual_seed(11)
    if kind == 'rows':        # 3x3 stride 1, 32-multiple width: row-reuse kernel forward / input gradient, 3x3 weight-gradient kernel
        return torch.randn([8, 128, 32, 32], generator=g), torch.randn([128, 128, 3, 3], generator=g) / 34, dict(padding=1)
    if kind == 'base':        # width 24: the base forward kernel; fp32 weight-gradient kernel
        return torch.randn([8, 64, 24, 24], generator=g), torch.randn([160, 64, 3, 3], generator=g) / 24, dict(padding=1)
    if kind == 'stride2':     # base forward kernel on a stride-2 lattice, stride-2 weight-gradient kernel
        return torch.randn([8, 64, 65, 65], generator=g), torch.randn([128, 64, 3, 3], generator=g) / 24, dict(stride=2)
    if kind == 'pointwise':   # 1x1: base forward kernel, 1x1 weight-gradient kernel
        return torch.randn([8, 192, 32, 32], generator=g), torch.randn([128, 192, 1, 1], generator=g) / 14, dict()
    raise KeyError(kind)


@pytest.mark.parametrize('kind', ['rows', 'base', 'stride2', 'pointwise'])
@pytest.mark.parametrize('mode', ['bf16x3', 'bf16'])
def test_reduced_product_modes(kind, mode):
    from torch_utils.ops import conv2d_gradfix as cg
    x, w, kw = _conv_case(kind)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, **kw)
    dy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(12))
    rx, rw = torch.autograd.grad(y64, [x64, w64], dy.double())
    old = cg.conv_math
    cg.conv_math = mode
    try:
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        y = cg.conv2d(xg, wg, **kw)
        gx, gw = torch.autograd.grad(y, [xg, wg], dy.cuda())
    finally:
        cg.conv_math = old
    hi, lo = REDUCED[mode]
    for name, got, ref in [('y', y, y64.detach()), ('dx', gx, rx), ('dw', gw, rw)]:
        err = float((got.detach().double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(kind, mode, name, f'{err:.2e}')
        assert err < hi, (kind, mode, name, err)
        if not (kind == 'base' and name in ('dx', 'dw')):      # those two run on fp32 kernels in every mode (64 output channels on few pixels; 24-pixel rows)
            assert err > lo, (kind, mode, name, err, 'suspiciously exact: did the mode reach the kernel?')


def test_plan_reports_the_reduced_modes():
    import ctypes
    from torch_utils import custom_ops
    from torch_utils.ops import _native
    lib = _native.lib()
    for mode, code in [('bf16x6', 2), ('bf16x3', 3), ('bf16', 4), ('f16x3', 5)]:
        d = custom_ops.ConvDesc(N=16, C_in=128, H=128, W=128, C_out=128, OH=128, OW=128, kh=3, kw=3, stride=1, pad_h=1, pad_w=1,
                                groups=1, transposed=0, flip=0, math=code, wscale=1.0)
        tile, ks, math, launches, kernel = (ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int())
        _native.check(lib.pasta_conv2d_plan(ctypes.byref(d), 0, ctypes.byref(tile), ctypes.byref(ks), ctypes.byref(math),
                                            ctypes.byref(launches), ctypes.byref(kernel)))
        # the row-reuse family in every split mode: 2-D tiles (eight waves for the six-product arithmetic: 7, else four-row tiles: 4), or the row kernel (2) under PASTA_ROWS2D=0
        import os
        assert math.value == code and kernel.value == (2 if os.environ.get('PASTA_ROWS2D') == '0' else 7 if mode in ('bf16x6', 'f16x3') else 4)


# ---- edge of the split-bf16 operand range ------------------------------------------------------------------------------
# v = v1 + v2 + v3 with bf16 pieces keeps fp32's exponent range (bf16 has the same 8 exponent bits), so large and small
# NORMAL operands behave like fp32.  Two deliberate differences from an fp32 FMA chain, both outside the model's operating
# range (activations are clamped to +-256 after every layer, networks.py:176-178, and gradients pass nan_to_num before the
# optimiser, training_loop_wo_flow_fullbody.py:513-515):
#  * an infinite operand gives NaN where fp32 gives +-inf: the second piece is bf16(inf - inf).  Outputs whose receptive field
#    does not contain the infinite element are unaffected;
#  * operands below ~2^-110 lose their low pieces to bf16 subnormals (flushed), i.e. the result keeps 8..16 significant bits
#    instead of 24 -- at magnitudes where the products themselves are at the edge of fp32's range.

@pytest.mark.parametrize('scale', [1e30, 1e-30])
def test_split_bf16_at_large_and_small_normal_magnitudes(scale):
    g = torch.Generator().manual_seed(3)
    x = torch.randn([4, 64, 64, 64], generator=g) * scale           # 16384 pixels, 128 output channels: the matrix-core tiles of every mode
    w = torch.randn([128, 64, 3, 3], generator=g) / 24
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    e16 = _errs('bf16x6', x.cuda(), w.cuda(), ref64)
    e32 = _errs('f32', x.cuda(), w.cuda(), ref64)
    eh3 = _errs('f16x3', x.cuda(), w.cuda(), ref64)               # the power-of-two operand scale brings any normal magnitude into fp16's range
    print(f'scale {scale}: split-bf16 {e16[0]:.3e}, fp16 x 3 {eh3[0]:.3e}, fp32 MFMA {e32[0]:.3e}')
    assert e16[0] < 5e-6 and e32[0] < 5e-6 and eh3[0] < 5e-6


def test_f16x3_dynamic_range_within_one_tensor():
    """PASTA_MATH_F16X3 scales an operand by ONE power of two per tensor.  Activations keep fp32-class relative accuracy down
    to 2^-28 of the tensor's largest element (pre-scaled low piece): an image region 1e-5 times smaller than another region of the same
    tensor comes out as accurately as the large one, with isolated elements 100 x the rest in the same tensor (largest / typical
    small element = 4e7 = 2^25).  Against fp64, region by region."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(6)
    x = torch.randn([2, 64, 64, 64], generator=g)
    x[:, :, :, 32:] *= 1e-5                                          # right half of every plane: 1e-5 of the left half
    x[0, ::5, ::7, ::3] *= 100                                       # and outliers in sample 0
    w = torch.randn([128, 64, 3, 3], generator=g) / 24
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    for mode in ['f16x3', 'f32']:
        cg.conv_math = mode
        try:
            y = cg.conv2d(x.cuda(), w.cuda(), padding=1).double().cpu()
        finally:
            cg.conv_math = 'default'
        for name, sl in [('large half', (slice(1, 2), slice(None), slice(None), slice(0, 30))), ('small half', (slice(1, 2), slice(None), slice(None), slice(34, 64)))]:
            d, r = (y[sl] - ref[sl]), ref[sl]
            rms = float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
            print(mode, name, f'{rms:.2e}')
            assert rms < 1e-6, (mode, name, rms)


def test_split_bf16_subnormal_pieces_lose_precision_gracefully():
    g = torch.Generator().manual_seed(4)
    x = torch.randn([4, 64, 32, 32], generator=g) * 1e-36
    w = torch.randn([64, 64, 3, 3], generator=g) / 24
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    e16 = _errs('bf16x6', x.cuda(), w.cuda(), ref64)
    print(f'scale 1e-36: split-bf16 max-rel {e16[0]:.3e}')
    assert e16[0] < 2e-2                # at least the leading bf16 piece survives: no garbage, no NaN
    eh3 = _errs('f16x3', x.cuda(), w.cuda(), ref64)
    print(f'scale 1e-36: fp16 x 3 max-rel {eh3[0]:.3e}')
    assert eh3[0] < 5e-6                # the scale (up to 2^120) lifts 1e-36 into the normal range: full accuracy


@pytest.mark.parametrize('mode', ['bf16x6', 'f16x3'])
@pytest.mark.parametrize('bad', [float('inf'), float('-inf'), float('nan')])
def test_non_finite_operand_stays_local(bad, mode):
    """One non-finite input element: every output whose 3x3 window contains it is non-finite (fp32 gives inf or NaN there as
    well), every other output is exactly what it is without the poisoned element."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(5)
    x = torch.randn([2, 64, 64, 64], generator=g).cuda()          # 8192+ pixels x 128 channels: the matrix-core tile of the mode
    w = (torch.randn([128, 64, 3, 3], generator=g) / 24).cuda()
    cg.conv_math = mode
    try:
        clean = cg.conv2d(x, w, padding=1)
        xp = x.clone()
        xp[1, 7, 10, 20] = bad
        y = cg.conv2d(xp, w, padding=1)         # (f16x3: the operand scale skips non-finite elements, so it is the clean tensor's)
    finally:
        cg.conv_math = 'default'
    hit = torch.zeros_like(y, dtype=torch.bool)
    hit[1, :, 9:12, 19:22] = True
    assert not torch.isfinite(y[hit]).any()
    assert torch.equal(y[~hit], clean[~hit])
    ref = torch.nn.functional.conv2d(xp.cpu(), w.cpu(), padding=1)
    assert not torch.isfinite(ref[hit.cpu()]).any()          # the fp32 reference is non-finite at the same outputs


@pytest.mark.parametrize('ratio,bound', [(1e-4, 1e-6), (1e-7, 2e-3)])
def test_f16x3_weight_gradient_dynamic_range(ratio, bound):
    """Both operands of a weight gradient carry ONE scale per tensor and keep fp32-class accuracy for elements within 2^-17
    (7.6e-6) of their tensor's largest.  Here dy lives only where x is `ratio` times smaller than elsewhere in the same tensor, so
    every product of the sum has a small x: at 1e-4 the gradient is as accurate as with uniform magnitudes (rms <= 1e-6 against
    fp64); at 1e-7, outside the range, the low pieces fall below fp16's normal numbers and accuracy degrades gracefully (an
    absolute error of 2^-39 of the tensor's maximum per element: ~1e-4 relative here), it does not collapse."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(8)
    x = torch.randn([4, 64, 64, 64], generator=g)
    x[:, :, :, 32:] *= ratio
    w = torch.randn([128, 64, 3, 3], generator=g) / 24
    dy = torch.randn([4, 128, 64, 64], generator=g)
    dy[:, :, :, :34] = 0                               # only the small half of x meets a gradient
    w64 = w.double().requires_grad_(True)
    rw, = torch.autograd.grad(torch.nn.functional.conv2d(x.double(), w64, padding=1), w64, dy.double())
    cg.conv_math = 'f16x3'
    try:
        wg = w.cuda().requires_grad_(True)
        gw, = torch.autograd.grad(cg.conv2d(x.cuda(), wg, padding=1), wg, dy.cuda())
    finally:
        cg.conv_math = 'default'
    rms = float((gw.double().cpu() - rw).pow(2).mean().sqrt() / rw.pow(2).mean().sqrt())
    print(f'ratio {ratio}: rms {rms:.2e}')
    assert rms < bound, rms
