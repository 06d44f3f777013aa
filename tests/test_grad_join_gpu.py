"""Gradients of multi-consumer tensors joined in a launch's epilogue (``passthrough`` of conv2d_gradfix._ConvBiasActHip, networks._GRAD_JOIN):
a residual block's ``x`` feeds conv0 AND the skip branch (reference networks.py:528-558, :959-997); autograd adds the two input gradients with a
pass of its own, here the skip branch's gradient is the residual operand of conv0's input-gradient launch.  Same values as the plain graph,
first and second derivatives."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.mark.parametrize('n,ci,co,hw,act', [(4, 64, 64, 32, 'lrelu'), (3, 128, 64, 64, 'linear'), (2, 20, 36, 17, 'relu'), (9, 64, 128, 32, 'lrelu')])
def test_passthrough_joins_the_other_consumers_gradient(n, ci, co, hw, act):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(n + ci + co)
    x = torch.randn([n, ci, hw, hw], generator=g).cuda()
    w = (torch.randn([co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)).cuda()
    b = torch.randn([co], generator=g).cuda()
    w2 = (torch.randn([co, ci, 1, 1], generator=g) / ci ** 0.5).cuda()
    dy = torch.randn([n, co, hw, hw], generator=g).cuda()

    def run(join):
        xs, ws, bs, w2s = (t.clone().requires_grad_(True) for t in (x, w, b, w2))
        h = xs * 1.5                                     # a non-leaf, as inside a network
        if join:
            y, hp = cg.conv2d_bias_act(h, ws, bs, padding=1, act=act, passthrough=True)
        else:
            y, hp = cg.conv2d_bias_act(h, ws, bs, padding=1, act=act), h
        z = y + cg.conv2d(hp, w2s)
        first = torch.autograd.grad(z, [xs, ws, bs, w2s], dy, create_graph=True)
        # R1-like second derivative: the squared norm of the input gradient, differentiated with respect to the parameters
        second = torch.autograd.grad(first[0].square().sum(), [ws, w2s], allow_unused=True)
        return (z,) + tuple(first) + tuple(second)
    got, want = run(True), run(False)
    for i, (u, v) in enumerate(zip(got, want)):
        assert (u is None) == (v is None)
        if u is not None:
            assert u.shape == v.shape and _rel(u, v) < 2e-6, (i, _rel(u, v))


def test_passthrough_alone_and_unused():
    """Only one of the two outputs differentiated: the other's gradient arrives as None."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(3)
    x = torch.randn([2, 32, 16, 16], generator=g).cuda().requires_grad_(True)
    w = torch.randn([32, 32, 3, 3], generator=g).cuda().requires_grad_(True)
    h = x * 2
    y, hp = cg.conv2d_bias_act(h, w, None, padding=1, passthrough=True)
    gx, = torch.autograd.grad(hp.sum(), [x], retain_graph=True)
    assert torch.equal(gx, torch.full_like(gx, 2.0))
    gx2, gw = torch.autograd.grad(y.sum(), [x, w])
    ref = cg.conv2d_bias_act(x * 2, w, None, padding=1)
    rx, rw = torch.autograd.grad(ref.sum(), [x, w])
    assert _rel(gx2, rx) < 1e-6 and _rel(gw, rw) < 1e-6
    with torch.no_grad():
        y2, hp2 = cg.conv2d_bias_act(h, w, None, padding=1, passthrough=True)
    assert hp2 is h and torch.equal(y2, y)


@pytest.mark.parametrize('block', ['res', 'res_down', 'disc'])
def test_blocks_equal_the_plain_graph(block):
    from training import networks
    g = torch.Generator().manual_seed(11)
    if block == 'disc':
        net = networks.DiscriminatorBlock(64, 64, 128, resolution=32, img_channels=3, first_layer_idx=0, architecture='resnet', conv_clamp=256).cuda()
        x = torch.randn([6, 64, 32, 32], generator=g).cuda()
        call = lambda m, t: m(t, None)[0]
    else:
        net = networks.ResBlock(64, 128 if block == 'res_down' else 64, kernel_size=4, activation='relu', down=2 if block == 'res_down' else 1).cuda()
        x = torch.randn([5, 64, 32, 32], generator=g).cuda()
        call = lambda m, t: m(t)
    params = list(net.parameters())

    def run(join):
        old, networks._GRAD_JOIN = networks._GRAD_JOIN, join
        try:
            xs = x.clone().requires_grad_(True)
            y = call(net, xs * 0.5)
            first = torch.autograd.grad(y.square().sum(), [xs] + params, create_graph=True)
            second = torch.autograd.grad(first[0].square().sum(), params, allow_unused=True)
            return (y,) + tuple(first) + tuple(second)
        finally:
            networks._GRAD_JOIN = old
    got, want = run(True), run(False)
    for i, (u, v) in enumerate(zip(got, want)):
        assert (u is None) == (v is None), i
        if u is not None:
            assert _rel(u, v) < 5e-6, (i, _rel(u, v))


def test_spade_blocks_chain_the_feature_map():
    from training import networks
    g = torch.Generator().manual_seed(2)
    blocks = [networks.Spade_ResBlockV2(32, 32, resolution=16, feat_channels=24).cuda() for _ in range(3)]
    x = torch.randn([3, 32, 16, 16], generator=g).cuda()
    feat = torch.randn([3, 24, 16, 16], generator=g).cuda()
    params = [p for b in blocks for p in b.parameters()]

    def run(join):
        old, networks._GRAD_JOIN = networks._GRAD_JOIN, join
        try:
            xs, fs = x.clone().requires_grad_(True), feat.clone().requires_grad_(True)
            h, f = xs * 1.0, fs * 1.0
            for b in blocks:
                h, f = b(h, f, return_feat=True)
            return (h,) + torch.autograd.grad(h.square().sum(), [xs, fs] + params)
        finally:
            networks._GRAD_JOIN = old
    got, want = run(True), run(False)
    for i, (u, v) in enumerate(zip(got, want)):
        assert _rel(u, v) < 5e-6, (i, _rel(u, v))


def test_spade_block_backward_twice_over_a_retained_graph():
    """the three normalisations of a block write dgamma | dbeta into ONE shared buffer that the split node hands out as the
    gradient; a second backward pass over the same graph (retain_graph=True, or autograd.grad followed by backward) must find the node's
    holder again and must get a FRESH buffer -- not overwrite the tensor the first pass returned."""
    from training import networks
    g = torch.Generator().manual_seed(4)
    blk = networks.Spade_ResBlockV2(32, 32, resolution=16, feat_channels=24).cuda()
    x = torch.randn([2, 32, 16, 16], generator=g).cuda().requires_grad_(True)
    feat = torch.randn([2, 24, 16, 16], generator=g).cuda().requires_grad_(True)
    params = list(blk.parameters())
    y = blk(x * 1.0, feat * 1.0)
    first = torch.autograd.grad(y.square().sum(), [x, feat] + params, retain_graph=True)
    kept = [t.clone() for t in first]
    second = torch.autograd.grad((2 * y).square().sum(), [x, feat] + params)             # another loss over the SAME graph: 4 x the gradients
    for i, (a, k, b) in enumerate(zip(first, kept, second)):
        assert torch.equal(a, k), i                                                   # what the first pass returned was not written again
        assert _rel(b, 4 * k) < 5e-6, (i, _rel(b, 4 * k))


def test_layer_sum_with_passthrough():
    """``Conv2dLayer.forward(add=..., passthrough=True)`` returns ``(layer(x) + add, x')``, fused or not."""
    from training import networks
    g = torch.Generator().manual_seed(6)
    x = torch.randn([2, 16, 16, 16], generator=g).cuda().requires_grad_(True)
    for kwargs in (dict(bias=False), dict(bias=True)):      # (fused in the epilogue; `add_` on the layer's output, as the reference writes it: linear layers only -- an activation's saved output must not be written)
        layer = networks.Conv2dLayer(16, 32, kernel_size=1, **kwargs).cuda()
        add = torch.randn([2, 32, 16, 16], generator=g).cuda()
        want = layer(x) + add
        y, again = layer(x, passthrough=True, add=add.clone())
        assert _rel(y, want) < 1e-6 and again.shape == x.shape
        gx, = torch.autograd.grad(y.sum() + again.sum(), [x])
        rx, = torch.autograd.grad(want.sum() + x.sum(), [x])
        assert _rel(gx, rx) < 5e-6


@pytest.mark.parametrize('shape,kw', [
    ([3, 16, 64, 64], dict(padding=[2, 2, 2, 2])),                      # blur in front of a stride-2 convolution: 64 -> 65 (odd pitch)
    ([2, 8, 65, 65], dict(padding=[1, 1, 1, 1], gain=4)),               # ... and behind a stride-2 conv_transpose2d
    ([2, 8, 64, 64], dict(down=2, padding=[1, 1, 1, 1])),               # the skip branch's decimation
    ([2, 8, 32, 32], dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    ([4, 512, 8, 8], dict(padding=[2, 2, 2, 2])),                       # small-plane kernel
])
def test_upfirdn2d_addend_and_passthrough(shape, kw):
    """``pasta_upfirdn2d(y_add)``: the filter's result plus a tensor in the one launch; and the pass-through form whose backward uses it."""
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(sum(shape))
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device='cuda')
    x = torch.randn(shape, generator=g).cuda()
    y0 = upfirdn2d.upfirdn2d(x, f, **kw)
    add = torch.randn(y0.shape, generator=g).cuda()
    up, down = kw.get('up', 1), kw.get('down', 1)
    px0, px1, py0, py1 = kw['padding']
    cfg = (up, up, down, down, px0, px1, py0, py1, False, kw.get('gain', 1))
    y1 = upfirdn2d._Upfirdn2dHip.apply(x, f, cfg, add)
    assert _rel(y1, y0 + add) < 1e-6

    w2 = torch.randn(shape[1], generator=g).cuda()
    def run(join):
        xs = x.clone().requires_grad_(True)
        h = xs * 1.25
        if join:
            y, hp = upfirdn2d.upfirdn2d(h, f, passthrough=True, **kw)
        else:
            y, hp = upfirdn2d.upfirdn2d(h, f, **kw), h
        z = y.square().sum() + (hp * w2[None, :, None, None]).square().sum()
        first, = torch.autograd.grad(z, [xs], create_graph=True)
        second, = torch.autograd.grad(first.square().sum(), [xs])
        return first, second
    for u, v in zip(run(True), run(False)):
        assert _rel(u, v) < 2e-6


def test_feature_pyramid_levels_are_handed_on():
    """StyleEncoderNetworkV16: a pyramid level feeds the next stride-2 layer and (later) the synthesis blocks' merge layers."""
    from training import networks
    g = torch.Generator().manual_seed(4)
    enc = networks.StyleEncoderNetworkV16(input_nc=6, output_nc=128, ngf=16, n_downsampling=4, feat_levels=3).cuda()
    x = torch.randn([2, 6, 64, 64], generator=g).cuda()
    c = torch.randn([2, 3, 64, 64], generator=g).cuda()
    params = list(enc.feat_enc.parameters())

    def run(join):
        old, networks._GRAD_JOIN = networks._GRAD_JOIN, join
        try:
            cs = c.clone().requires_grad_(True)
            code, pyramid = enc(x, cs)
            loss = sum((lvl * (i + 1.5)).square().sum() for i, lvl in enumerate(pyramid))
            return tuple(pyramid) + torch.autograd.grad(loss, [cs] + params)
        finally:
            networks._GRAD_JOIN = old
    for i, (u, v) in enumerate(zip(run(True), run(False))):
        assert _rel(u, v) < 5e-6, (i, _rel(u, v))
