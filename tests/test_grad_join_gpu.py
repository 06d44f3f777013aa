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
