"""Forward-only modulated convolution (styles in the convolution's staging / weight packing, demodulation + noise + bias +
activation in its epilogue; reference training/networks.py:36-94, 263-315) against the differentiable path of the same
layers -- which the reference-written fixtures of tests/test_layers.py and tests/test_models_gpu.py pin -- and the
demodulation-coefficient kernel against the reference formula (networks.py:65-68)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda', 0)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('n,o,i,k', [(4, 64, 64, 3), (16, 512, 512, 3), (3, 7, 33, 1), (2, 96, 200, 3)])
def test_demod_coefs_match_the_reference_formula(n, o, i, k):
    from torch_utils.ops import conv2d_gradfix
    g = torch.Generator(device='cpu').manual_seed(n * 1000 + o)
    w = torch.randn([o, i, k, k], generator=g).to(_dev()).requires_grad_(True)
    s = (torch.randn([n, i], generator=g) * 1.5).to(_dev()).requires_grad_(True)
    d = conv2d_gradfix.demod_coefs(w, s)
    # networks.py:65-68 in fp64
    w64, s64 = w.detach().double().requires_grad_(True), s.detach().double().requires_grad_(True)
    ref = ((w64[None] * s64[:, None, :, None, None]).square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    assert _rel(d, ref) < 2e-6
    probe = torch.randn(d.shape, generator=g).to(_dev())
    dw, ds = torch.autograd.grad((d * probe).sum(), [w, s], create_graph=True)
    rw, rs = torch.autograd.grad((ref * probe.double()).sum(), [w64, s64], create_graph=True)
    assert _rel(dw, rw) < 1e-5 and _rel(ds, rs) < 1e-5
    # second order (the path-length regulariser differentiates the styles' gradient again)
    ddw, = torch.autograd.grad(ds.square().sum(), [w])
    rdw, = torch.autograd.grad(rs.square().sum(), [w64])
    assert _rel(ddw, rdw) < 1e-4


def _layer(cls, **kw):
    torch.manual_seed(0)
    layer = cls(**kw).to(_dev())
    with torch.no_grad():
        for name, prm in layer.named_parameters():
            if name.endswith('bias'):
                prm.copy_(torch.randn_like(prm) * 0.3 + (1.0 if 'affine' in name else 0.0))
            if name == 'noise_strength':
                prm.fill_(0.37)
    return layer


@pytest.mark.parametrize('cin,cout,res,up', [(64, 64, 64, 1), (128, 64, 64, 2), (512, 512, 16, 1), (512, 256, 32, 2), (48, 40, 32, 1),
                                             (128, 128, 256, 2)])      # 128^2 -> 257^2 with per-sample weights: parity-pair kernel + the fp32 remainder kernel on modulated weights
@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('noise_mode', ['const', 'none'])
def test_synthesis_layer_forward_only_equals_the_differentiable_path(cin, cout, res, up, fused, noise_mode):
    from training.networks import SynthesisLayer
    layer = _layer(SynthesisLayer, in_channels=cin, out_channels=cout, w_dim=64, resolution=res, up=up, conv_clamp=256)
    g = torch.Generator(device='cpu').manual_seed(res + cin)
    x = torch.randn([4, cin, res // up, res // up], generator=g).to(_dev())
    w = torch.randn([4, 64], generator=g).to(_dev())
    ref = layer(x.clone().requires_grad_(True), w, noise_mode=noise_mode, fused_modconv=fused, gain=0.7).detach()     # records a graph: the training kernels
    with torch.no_grad():
        out = layer(x, w, noise_mode=noise_mode, fused_modconv=fused, gain=0.7)
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert _rel(out, ref) < 2e-5, _rel(out, ref)


def test_random_noise_is_added_per_sample_in_the_epilogue():
    from training.networks import SynthesisLayer
    layer = _layer(SynthesisLayer, in_channels=64, out_channels=64, w_dim=64, resolution=32, conv_clamp=256)
    x = torch.randn([4, 64, 32, 32], device=_dev())
    w = torch.randn([4, 64], device=_dev())
    torch.manual_seed(5)
    ref = layer(x.clone().requires_grad_(True), w, noise_mode='random', fused_modconv=False).detach()
    torch.manual_seed(5)
    with torch.no_grad():
        out = layer(x, w, noise_mode='random', fused_modconv=False)
    assert _rel(out, ref) < 2e-5


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('up', [1, 2])
def test_forward_only_layer_in_16_bit_storage(dtype, up):
    from training.networks import SynthesisLayer
    layer = _layer(SynthesisLayer, in_channels=64, out_channels=64, w_dim=64, resolution=64, up=up, conv_clamp=256)
    x = torch.randn([4, 64, 64 // up, 64 // up], device=_dev())
    w = torch.randn([4, 64], device=_dev())
    with torch.no_grad():
        ref = layer(x, w, noise_mode='const', fused_modconv=False)
        out = layer(x.to(dtype), w, noise_mode='const', fused_modconv=False)
    assert out.dtype == dtype
    assert _rel(out.float(), ref) < (2e-2 if dtype == torch.bfloat16 else 4e-3)


@pytest.mark.parametrize('fused', [False, True])
def test_torgb_heads_forward_only(fused):
    from training.networks import ToRGBLayer, ToRGBLayerFull
    for cls, kw in [(ToRGBLayer, {}), (ToRGBLayerFull, {})]:
        layer = _layer(cls, in_channels=64, out_channels=3, w_dim=64, conv_clamp=256, **kw)
        x = torch.randn([4, 64, 64, 64], device=_dev()) * 3
        w = torch.randn([4, 64], device=_dev())
        ref = layer(x.clone().requires_grad_(True), w, fused_modconv=fused)
        with torch.no_grad():
            out = layer(x, w, fused_modconv=fused)
        ref = ref if isinstance(ref, (tuple, list)) else [ref]
        out = out if isinstance(out, (tuple, list)) else [out]
        assert len(ref) == len(out)
        for a, b in zip(out, ref):
            assert (a is None) == (b is None)
            if a is not None:
                assert _rel(a, b.detach()) < 2e-5


def test_forward_only_refuses_to_drop_a_graph():
    from torch_utils.ops import conv2d_gradfix
    x = torch.randn([2, 16, 8, 8], device=_dev(), requires_grad=True)
    w = torch.randn([16, 16, 3, 3], device=_dev())
    s = torch.randn([2, 16], device=_dev())
    with pytest.raises(AssertionError):
        conv2d_gradfix.modulated_conv2d_forward(x, w, s, padding=1)
