"""One packing launch for a convolution and its input gradient (round 5; include/pasta_hip.h: pasta_conv2d_pack_pair, pasta_conv_desc.w_prepacked).
The forward of a convolution whose input needs a gradient packs both orientations of its weight; the backward's input-gradient launch finds its
workspace packed.  Nothing about the arithmetic changes: every result is bit-identical to the launches that pack for themselves."""

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,ci,co,h,k,stride,transposed', [
    (9, 32, 64, 32, 3, 1, False),        # 2-D tiles forward, input gradient on the same kernels
    (2, 32, 64, 32, 3, 1, False),        # a small lattice with 64 output channels: fp32-MFMA tiles, no pair
    (2, 64, 128, 33, 3, 2, False),       # stride 2: the input gradient is a transposed convolution (parity classes)
    (1, 128, 64, 64, 3, 2, True),        # the up path: forward on the one-pass kernel, input gradient on the stride-2 forward kernel
    (2, 48, 40, 20, 1, 1, False),        # pointwise, channel tails
    (2, 6, 64, 64, 1, 1, False),         # few-channel launches pack nothing: the pair is refused, the launches run as always
    (3, 512, 512, 4, 3, 1, False),       # K-sliced small planes
])
def test_pair_packing_changes_no_bit(n, ci, co, h, k, stride, transposed):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(n + ci + co + h)
    x0 = torch.randn([n, ci, h, h], generator=g).cuda()
    w0 = (torch.randn([ci, co, k, k] if transposed else [co, ci, k, k], generator=g) * 0.1).cuda()
    op = cg.conv_transpose2d if transposed else cg.conv2d
    calls = []
    real = cg._pack_pair

    def counted(x, w, cfg):
        r = real(x, w, cfg)
        calls.append(r[0] is not None)
        return r

    def run(pair):
        old, cg._PACK_PAIR, cg._pack_pair = cg._PACK_PAIR, pair, counted
        try:
            x = x0.clone().requires_grad_(True)
            w = w0.clone().requires_grad_(True)
            y = op(x, w, stride=stride, padding=k // 2 if not transposed else 0)
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda()
            gx, gw = torch.autograd.grad(y, [x, w], dy)
        finally:
            cg._PACK_PAIR, cg._pack_pair = old, real
        return y, gx, gw

    a = run(False)
    assert calls and not any(calls)
    del calls[:]
    b = run(True)
    # one pair per forward -- refused where one of the two launches packs differently or not at all (few-channel kernels, the fp32-MFMA tiles of
    # small lattices with few output channels, the packed-K stems): what the planner says of the two descriptors
    import ctypes
    from torch_utils.ops import _native
    cfg = cg._Cfg((transposed, stride, 0 if transposed else k // 2, 0 if transposed else k // 2, 0, 0, 1, 1.0))
    oh, ow = cg._out_hw(cfg, h, h, k, k)
    gcfg = cg._grad_cfg(cfg, (h, h), (oh, ow), k, k)
    ok = True
    for d in (cg._desc(cfg, (n, ci, h, h), co, oh, ow, k, k), cg._desc(gcfg, (n, co, oh, ow), ci, h, h, k, k)):
        kern, math = ctypes.c_int(), ctypes.c_int()
        assert _native.lib().pasta_conv2d_plan(ctypes.byref(d), 0, None, None, ctypes.byref(math), None, ctypes.byref(kern)) == 0
        ok = ok and math.value == cg.MATH_CODES['f16x3'] and kern.value not in (8, 11, 12)
    assert calls == [ok]
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_layer_with_epilogue_and_second_derivative():
    """conv2d_bias_act (the fused layer) with joined pass-through, first derivatives with a graph, then a second derivative: the nested launches of
    the recorded backward pack for themselves, the results equal the unpaired run bit for bit."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn([2, 32, 32, 32], generator=g).cuda()
    w0 = (torch.randn([64, 32, 3, 3], generator=g) * 0.1).cuda()
    b0 = torch.randn([64], generator=g).cuda()

    def run(pair):
        old, cg._PACK_PAIR = cg._PACK_PAIR, pair
        try:
            x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            y = cg.conv2d_bias_act(x, w, b, padding=1, act='lrelu')
            (gx,) = torch.autograd.grad(y.square().sum(), [x], create_graph=True)
            (ggw,) = torch.autograd.grad(gx.square().sum(), [w])
        finally:
            cg._PACK_PAIR = old
        return y.detach(), gx.detach(), ggw

    for u, v in zip(run(False), run(True)):
        assert torch.equal(u, v)
