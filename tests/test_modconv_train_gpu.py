"""The training path's shared-weight modulated convolution without the tensor x * styles (conv2d_gradfix.modulated_conv2d_shared; reference
training/networks.py:72-76): styles in the staging of the forward launch, in the epilogue of the input gradient, in the reduction of the
weight gradient (pasta_conv2d_wgrad_modulated).  Against torch's fp64 convolution of x * s."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.mark.parametrize('n,ci,co,hw,k', [
    (4, 64, 64, 64, 3),            # one 64 x 64 tile of (a, b); K slices rounded up to a multiple of 4
    (16, 128, 128, 32, 3),         # batch of the training step
    (3, 96, 160, 32, 3),           # channel tails in both tile directions, a batch that is not a power of two
    (2, 512, 512, 32, 3),
    (5, 64, 3, 64, 1),             # ToRGB: pointwise, three output channels
    (8, 64, 9, 32, 1),             # ToRGB with extra heads
    (4, 256, 256, 64, 1),
])
def test_values_and_all_three_gradients(n, ci, co, hw, k):
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(n * 7 + ci + co + k)
    x = torch.randn([n, ci, hw, hw], generator=g)
    w = torch.randn([co, ci, k, k], generator=g) / (k * ci ** 0.5)
    s = torch.randn([n, ci], generator=g) * 0.5 + 1.0
    s[:, ::7] *= 4.0                                          # styles of different sizes per channel and sample
    dy = torch.randn([n, co, hw, hw], generator=g)
    xr, wr, sr = (t.double().requires_grad_(True) for t in (x, w, s))
    yr = torch.nn.functional.conv2d(xr * sr[:, :, None, None], wr, padding=k // 2)
    want = (yr,) + torch.autograd.grad(yr, [xr, wr, sr], dy.double())
    xc, wc, sc = (t.cuda().requires_grad_(True) for t in (x, w, s))
    assert cg.modconv_available(xc, wc, sc, padding=k // 2)
    y = cg.modulated_conv2d_shared(xc, wc, sc, padding=k // 2)
    got = (y,) + torch.autograd.grad(y, [xc, wc, sc], dy.cuda())
    for name, u, v in zip(['y', 'dx', 'dw', 'ds'], got, want):
        assert u.shape == v.shape, name
        assert _rel(u, v) < (2e-5 if name in ('dw', 'ds') else 5e-6), (name, _rel(u, v))


def test_shapes_without_a_sample_aligned_kernel_are_declined():
    from torch_utils.ops import conv2d_gradfix as cg
    x = torch.randn([4, 512, 8, 8]).cuda()          # rows of 8 pixels: the generic weight-gradient kernel
    w = torch.randn([512, 512, 3, 3]).cuda()
    s = torch.randn([4, 512]).cuda()
    assert not cg.modconv_available(x, w, s, padding=1)
    x = torch.randn([4, 6, 64, 64]).cuda()          # few input channels
    assert not cg.modconv_available(x, torch.randn([64, 6, 3, 3]).cuda(), torch.randn([4, 6]).cuda(), padding=1)


def test_synthesis_layer_equals_the_scale_planes_path():
    """SynthesisLayer / ToRGB in training mode with and without the new path: outputs and every parameter gradient."""
    from training import networks
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(9)
    layer = networks.SynthesisLayer(64, 64, w_dim=32, resolution=64).cuda()
    x = torch.randn([4, 64, 64, 64], generator=g).cuda()
    wl = torch.randn([4, 32], generator=g).cuda()
    dy = torch.randn([4, 64, 64, 64], generator=g).cuda()
    params = list(layer.parameters())

    def run(on):
        old, cg._MODCONV = cg._MODCONV, on
        try:
            xs = x.clone().requires_grad_(True)
            y = layer(xs * 1.0, wl, noise_mode='const', fused_modconv=False)
            return (y,) + torch.autograd.grad(y, [xs] + params, dy, allow_unused=True)
        finally:
            cg._MODCONV = old
    got, want = run(True), run(False)
    for i, (u, v) in enumerate(zip(got, want)):
        assert (u is None) == (v is None)
        if u is not None:
            assert _rel(u, v) < 2e-5, (i, _rel(u, v))
