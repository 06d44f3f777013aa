"""Producer-written operand pieces (round 5; csrc/pieces.hip, include/pasta_hip.h): the low-pass in front of a stride-2 convolution
(reference conv2d_resample.py:119-122) writes the fp16 pieces h | l' of the default arithmetic instead of an fp32 tensor, and the stride-2
forward convolution and its weight gradient copy them.

* the pieces ARE the split of pasta_upfirdn2d's fp32 values (22 of their 24 bits);
* forward convolution, weight gradient, the layer with its epilogue, first and second derivatives equal the fp32-tensor path and fp64;
* the operand scale is a power of two: halving / doubling the bound row changes no bit of a result;
* shapes the kernels do not cover keep the fp32 tensor.
"""

import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _filter():
    from torch_utils.ops import upfirdn2d
    return upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()


@pytest.mark.parametrize('shape,pad', [
    ([2, 16, 64, 64], (2, 2, 2, 2)),         # 65 x 65: the remainder column of the last tile
    ([1, 8, 40, 72], (2, 2, 2, 2)),          # 41 x 73: partial tiles in both directions
    ([2, 24, 17, 130], (1, 1, 1, 1)),        # 16 x 129: pad 1, 129 = 2 x 64 + 1
    ([3, 8, 9, 9], (2, 2, 2, 2)),            # a plane smaller than one tile
    ([1, 32, 256, 256], (2, 2, 2, 2)),       # the live 256 -> 257 shape
])
def test_pieces_are_the_split_of_the_fp32_blur(shape, pad):
    from torch_utils.ops import conv2d_gradfix as cg, upfirdn2d
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).cuda()
    x[0, 0, 0, 0] = 37.5                                         # the largest magnitude sits in a corner: the bound is not the output's own maximum
    f = _filter()
    ref = upfirdn2d.upfirdn2d(x, f, padding=list(pad))
    pieces, bound, lshape = cg.blur_pieces(x, f, pad)
    assert tuple(lshape) == tuple(ref.shape) and pieces.numel() == ref.numel() * 4
    # the bound row: the input's partial maxima times gain * sum |f| (= 1 up to rounding), never below the output's largest magnitude
    assert abs(float(bound.max()) / float(x.abs().max()) - 1) < 1e-6 and float(bound.max()) * (1 + 1e-6) >= float(ref.abs().max())
    got = cg.pieces_unpack(pieces, bound, lshape)
    amax = float(bound.max())
    err = (got.double() - ref.double()).abs()
    # h + 2^-11 l' keeps 22 significand bits of v S: |error| <= 2^-23 |v| + one unit of l' at the tensor's scale (2^-11 x 2^-11 x 2^-14 amax S ...)
    assert float((err - ref.double().abs() * 2.0 ** -22).max()) <= amax * 2.0 ** -26, float(err.max())
    # the pieces themselves: h = fp16(v S), l' = fp16(2^11 (v S - h)), S the power of two that puts the bound into [2^13, 2^14)
    S = 2.0 ** (13 - torch.floor(torch.log2(bound.max())).item())
    assert 2 ** 13 <= amax * S < 2 ** 14
    n, c, oh, ow = lshape
    units = pieces.view(torch.float16).reshape(n, c // 8, oh, 2, ow, 8)            # [N][C / 8][H][piece][W] units of eight channels
    h = units[:, :, :, 0].permute(0, 1, 4, 2, 3).reshape(n, c, oh, ow)
    lp = units[:, :, :, 1].permute(0, 1, 4, 2, 3).reshape(n, c, oh, ow)
    vs = ref * S
    assert torch.equal(h, vs.half())
    assert torch.equal(lp, ((vs - vs.half().float()) * 2048.0).half())


def _down(x, w, b, f, pieces_on, **kw):
    from torch_utils.ops import conv2d_gradfix as cg, conv2d_resample
    old, cg._PIECES = cg._PIECES, pieces_on
    try:
        return conv2d_resample.conv2d_resample_bias_act(x=x, w=w, b=b, f=f, down=2, padding=1, **kw)
    finally:
        cg._PIECES = old


def _kernel_ids(n, ci, co, h):
    from torch_utils import custom_ops
    from torch_utils.ops import _native
    oh = (h - 3) // 2 + 1
    d = custom_ops.ConvDesc(N=n, C_in=ci, H=h, W=h, C_out=co, OH=oh, OW=oh, kh=3, kw=3, stride=2, pad_h=0, pad_w=0, groups=1, transposed=0, flip=0, math=0, x_layout=1)
    k, kw = ctypes.c_int(-1), ctypes.c_int(-1)
    rf = _native.lib().pasta_conv2d_plan(ctypes.byref(d), 4, None, None, None, None, ctypes.byref(k))
    rw = _native.lib().pasta_conv2d_wgrad_plan(ctypes.byref(d), ctypes.byref(kw))
    return (rf, k.value), (rw, kw.value)


@pytest.mark.parametrize('n,ci,co,h', [
    (4, 64, 128, 128),          # 129 -> 64: the 128-row tile
    (8, 32, 64, 64),            # 65 -> 32: the 64-row tile, two rows of outputs per pixel tile ... (8192 pixels exactly: not taken) -> see below
    (16, 32, 64, 64),           # 65 -> 32 over 16384 pixels: the 64-row tile
    (2, 24, 48, 256),           # 257 -> 128: a channel count that fills one and a half K chunks (24 = 16 + 8)
])
def test_down_layer_equals_the_fp32_tensor_path_and_fp64(n, ci, co, h):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(n + ci + co + h)
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    w = (torch.randn([co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)).cuda()
    b = torch.randn([co], generator=g).cuda()
    f = _filter()
    taken = cg.pieces_available(x, f, w, (2, 2, 2, 2))
    (rf, kf), (rw, kwg) = _kernel_ids(n, ci, co, h + 1)
    assert taken == (rf == 0 and kf == 10 and rw == 0 and kwg == 6)
    assert taken == (n * (h // 2) ** 2 > 8192)
    xs = [x.clone().requires_grad_(True) for _ in range(2)]
    ws = [w.clone().requires_grad_(True) for _ in range(2)]
    bs = [b.clone().requires_grad_(True) for _ in range(2)]
    outs = []
    for i, on in enumerate((True, False)):
        y = _down(xs[i], ws[i], bs[i], f, on, act='lrelu', gain=2 ** 0.5, clamp=256)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).cuda()
        outs.append((y,) + torch.autograd.grad(y, [xs[i], ws[i], bs[i]], dy))
    for a, r in zip(*outs):
        assert _rel(a, r) < (2e-6 if taken else 1e-12), _rel(a, r)
    # fp64: blur, stride-2 convolution, the slopes the GPU took
    x64, w64, b64 = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    f64 = f.double().cpu()
    xb = torch.nn.functional.conv2d(torch.nn.functional.pad(x64, [2, 2, 2, 2]).reshape(n * ci, 1, h + 4, h + 4), f64.flip([0, 1])[None, None]).reshape(n, ci, h + 1, h + 1)
    pre = torch.nn.functional.conv2d(xb, w64, stride=2) + b64.reshape(1, -1, 1, 1)
    y0 = outs[0][0].detach().cpu()
    y64 = (torch.where(y0 > 0, pre, pre * 0.2) * 2 ** 0.5).clamp(-256, 256)
    dy = torch.randn(y0.shape, generator=torch.Generator().manual_seed(5)).double()
    r = (y64,) + torch.autograd.grad(y64, [x64, w64, b64], dy)
    for a, rr, tol in zip(outs[0], r, (3e-6, 5e-6, 1e-5, 1e-5)):
        assert _rel(a, rr) < tol, (_rel(a, rr), tol)


def test_results_do_not_depend_on_which_power_of_two_scaled_the_operand():
    """the producer takes its scale from a BOUND of the output it has before it starts; any admissible power of two gives the
    same bits in the consumers (h, l' and every product scale exactly), as long as nothing overflows or leaves fp16's normal range."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(3)
    n, ci, co, h = 4, 64, 128, 128
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    w = (torch.randn([co, ci, 3, 3], generator=g) / 24).cuda()
    dy = torch.randn([n, co, 64, 64], generator=g).cuda()
    f = _filter()
    cfg = cg._Cfg((False, 2, 0, 0, 0, 0, 1, 1.0))
    parts = cg.tensor_amax(x)
    ys, dws = [], []
    for mul in (1.0, 2.0, 0.5, 4.0, 0.25):
        pieces, bound, shape = cg.blur_pieces(x, f, (2, 2, 2, 2), x_amax=parts * mul)
        assert torch.equal(bound, parts * mul * (bound.max() / (parts.max() * mul)))       # the same row, scaled
        ys.append(cg._launch_conv(pieces, w, cfg, pieces=(bound, shape)))
        dws.append(cg._launch_wgrad_pieces(pieces, dy, cfg, tuple(w.shape), (bound, shape)))
    for y, dw in zip(ys[1:], dws[1:]):
        assert torch.equal(y, ys[0]) and torch.equal(dw, dws[0])
    # and the values are right (fp64)
    xb = torch.nn.functional.conv2d(torch.nn.functional.pad(x.double().cpu(), [2, 2, 2, 2]).reshape(n * ci, 1, h + 4, h + 4),
                                    f.double().cpu().flip([0, 1])[None, None]).reshape(n, ci, h + 1, h + 1).requires_grad_(True)
    w64 = w.double().cpu().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(xb, w64, stride=2)
    dw64, = torch.autograd.grad(y64, [w64], dy.double().cpu())
    assert _rel(ys[0], y64) < 3e-6 and _rel(dws[0], dw64) < 1e-5


def test_weight_gradient_from_pieces_per_tap_and_ragged_channels():
    """The transposed LDS reads gather every second halo column for the nine taps: each tap's slice of dw against fp64, with channel counts that do
    not fill the 64 x 64 tile (Bg = 24: one and a half octets of the second 16-channel group idle; Ag = 48)."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(8)
    n, ci, co, h = 3, 24, 48, 130          # blurred plane 131 x 131 -> 65 x 65 ... rows of 65 pixels are not a multiple of 16: not taken
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    w = torch.randn([co, ci, 3, 3], generator=g).cuda()
    assert not cg.pieces_available(x, _filter(), w, (2, 2, 2, 2))
    n, ci, co, h = 5, 24, 48, 128          # 129 -> 64: taken (5 x 64 x 64 > 8192)
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    x[:, :, ::7, ::5] *= 30.0
    dy = torch.randn([n, co, 64, 64], generator=g).cuda()
    f = _filter()
    assert cg.pieces_available(x, f, w, (2, 2, 2, 2))
    cfg = cg._Cfg((False, 2, 0, 0, 0, 0, 1, 0.5))              # a weight gain rides in the reduction
    pieces, bound, shape = cg.blur_pieces(x, f, (2, 2, 2, 2))
    dw = cg._launch_wgrad_pieces(pieces, dy, cfg, tuple(w.shape), (bound, shape))
    xb = torch.nn.functional.conv2d(torch.nn.functional.pad(x.double().cpu(), [2, 2, 2, 2]).reshape(n * ci, 1, h + 4, h + 4),
                                    f.double().cpu().flip([0, 1])[None, None]).reshape(n, ci, h + 1, h + 1)
    w64 = w.double().cpu().requires_grad_(True)
    dw64, = torch.autograd.grad(torch.nn.functional.conv2d(xb, w64 * 0.5, stride=2), [w64], dy.double().cpu())
    for r in range(3):
        for c in range(3):
            assert _rel(dw[:, :, r, c], dw64[:, :, r, c]) < 1e-5, (r, c, _rel(dw[:, :, r, c], dw64[:, :, r, c]))


def test_second_derivatives_through_the_down_layer():
    """R1 (loss_wo_flow_fullbody.py:246-254) differentiates the input gradient of the discriminator: the backward of the fused function records a
    graph of differentiable operators, and the weight gradient of THAT pass comes from the recomputed fp32 blur."""
    from training import networks
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(12)
    layer = networks.Conv2dLayer(32, 64, kernel_size=3, activation='lrelu', down=2, conv_clamp=256).cuda()
    x0 = torch.randn([16, 32, 64, 64], generator=g).cuda()
    res = []
    for on in (True, False):
        old, cg._PIECES = cg._PIECES, on
        try:
            x = x0.clone().requires_grad_(True)
            assert cg.pieces_available(x, layer.resample_filter, layer.weight, (2, 2, 2, 2)) == on
            y = layer(x)
            with cg.no_weight_gradients():
                gx, = torch.autograd.grad(y.square().sum(), [x], create_graph=True)
            pen = gx.square().sum()
            gw, gb = torch.autograd.grad(pen, [layer.weight, layer.bias])
            first = torch.autograd.grad(layer(x).sum(), [layer.weight])        # a plain first-order weight gradient (the pieces kernel)
            res.append((y, gx, gw, gb) + first)
        finally:
            cg._PIECES = old
    for i, (a, r) in enumerate(zip(*res)):
        assert _rel(a, r) < 5e-6, (i, _rel(a, r))


def test_discriminator_block_with_joined_gradients_on_pieces():
    """The residual block's conv1 is a down layer behind conv0 (passthrough of the filter, residual sum in the skip convolution): same values with
    the blurred tensors as pieces or as fp32."""
    from training import networks
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(14)
    net = networks.DiscriminatorBlock(32, 32, 64, resolution=64, img_channels=3, first_layer_idx=0, architecture='resnet', conv_clamp=256).cuda()
    x0 = torch.randn([16, 32, 64, 64], generator=g).cuda()
    params = list(net.parameters())
    res = []
    for on in (True, False):
        old, cg._PIECES = cg._PIECES, on
        try:
            x = x0.clone().requires_grad_(True)
            y = net(x * 0.5, None)[0]
            res.append((y,) + torch.autograd.grad(y.square().sum(), [x] + params))
        finally:
            cg._PIECES = old
    for i, (a, r) in enumerate(zip(*res)):
        assert _rel(a, r) < 5e-6, (i, _rel(a, r))


@pytest.mark.parametrize('n,ci,co,h,transposed', [
    (2, 32, 128, 32, False),         # one tile per image
    (1, 40, 256, 64, False),         # an octet count that leaves half a chunk empty; two output tiles
    (2, 64, 128, 32, True),          # the input gradient of a 3x3 layer (conv_transpose2d, stride 1)
])
def test_stride_one_tile_kernel_reads_pieces_bit_for_bit(n, ci, co, h, transposed):
    """conv_fwd_rows2d_bf16x6_kernel<..., XP> (plan kernel 7 with x_layout = PASTA_LAYOUT_PIECES16): the same pieces the kernel forms in its
    staging, written by pasta_pieces_pack -- not one bit of the result moves; pack and unpack are inverse up to the 22 bits kept."""
    from torch_utils.ops import conv2d_gradfix as cg, _native
    g = torch.Generator().manual_seed(n + ci + co)
    x = torch.randn([n, ci, h, h], generator=g).cuda()
    w = (torch.randn([ci, co, 3, 3] if transposed else [co, ci, 3, 3], generator=g) * 0.1).cuda()
    cfg = cg._Cfg((transposed, 1, 1, 1, 0, 0, 1))
    _native.amax_attach(x, cg.tensor_amax(x))
    pieces, bound, shape = cg.pieces_pack(x)
    assert _rel(cg.pieces_unpack(pieces, bound, shape), x) < 2.0 ** -21
    y0 = cg._launch_conv(x, w, cfg)
    y1 = cg._launch_conv(pieces, w, cfg, pieces=(bound, shape))
    assert torch.equal(y0, y1)
    ref = (torch.nn.functional.conv_transpose2d if transposed else torch.nn.functional.conv2d)(x.double(), w.double(), padding=1)
    assert _rel(y1, ref) < 2e-6


def test_pieces_on_a_launch_no_kernel_takes_are_refused():
    from torch_utils.ops import conv2d_gradfix as cg
    x = torch.randn([2, 32, 16, 16]).cuda()              # a plane without 8 x 32 tiles: the base kernel's launch
    w = torch.randn([128, 32, 3, 3]).cuda()
    pieces, bound, shape = cg.pieces_pack(x)
    with pytest.raises(RuntimeError, match='PASTA_LAYOUT_PIECES16'):
        cg._launch_conv(pieces, w, cg._Cfg((False, 1, 1, 1, 0, 0, 1)), pieces=(bound, shape))


def test_one_operand_packed_once_for_several_three_by_three_layers():
    """share_pieces (the SPADE feature map, networks.py Spade_ResBlockV2._batched_gamma_beta): the tensor is packed once, every conv_mlp launch
    that reads it -- directly or through the pass-through output of the previous reader -- copies the pieces, and nothing moves: outputs and
    gradients are bit-identical to the launches that split the fp32 tensor themselves."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(21)
    x0 = torch.randn([2, 32, 32, 32], generator=g).cuda()
    ws = [(torch.randn([128, 32, 3, 3], generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(3)]
    dys = [torch.randn([2, 128, 32, 32], generator=g).cuda() for _ in range(3)]

    def run(share):
        seen = []
        def hook(kind, desc, launch, flags=0):
            seen.append((kind, int(desc.x_layout)))
            launch()
        x = (x0 * 1.0).requires_grad_(True)          # (a non-leaf with a graph, like the feature map)
        feat = x * 1.0
        if share:
            cg.share_pieces(feat)
        outs = []
        cg.launch_hook = hook
        try:
            for w in ws:
                y, feat = cg.conv2d_bias_act(feat, w, None, padding=1, act='relu', passthrough=True)
                outs.append(y)
        finally:
            cg.launch_hook = None
        grads = torch.autograd.grad(outs, [x] + ws, dys)
        return outs, grads, seen

    o0, g0, s0 = run(False)
    o1, g1, s1 = run(True)
    assert [k for k in s0 if k[0] == 'conv'] == [('conv', 0)] * 3 and [k for k in s1 if k[0] == 'conv'] == [('conv', 1)] * 3
    for a, b in zip(o0 + list(g0), o1 + list(g1)):
        assert torch.equal(a, b)
