"""Layer-level parity (SURVEY.md 8c 'Primitives'): Conv2dLayer, SynthesisLayer, ToRGBLayerFull, Spade_Norm_Block,
Spade_ResBlockV2, MinibatchStdLayer, DiscriminatorBlock on small channels, against fixtures written by the reference's
own classes (oracle/make_golden.py --only primitives -> tests/golden/layers_primitives.npz).

CPU: the oracle's functional restatement against the fixtures.  GPU: this repository's layer classes (HIP path,
through the C ABI) against the same fixtures."""

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import param_fill as PF
from oracle import primitive_cases as PC
from oracle import ref_networks as RN

TOL = 2e-5          # fp32 layers of a few thousand products per output; gradients summed over <= 2048 pixels
CASE_IDS = [c['name'] for c in PC.CASES]


def _state(case):
    """Closed-form weights of the case's layer as an oracle state dict with prefix 'L'."""
    from training import networks          # constructing a layer needs no GPU
    layer = PF.fill_module(getattr(networks, case['cls'])(**case['ctor']))
    sd = {'L.' + k: v.detach().clone() for k, v in layer.state_dict().items()}
    return layer, sd


def _oracle_forward(case, sd, xs):
    c, f = case['ctor'], case['fwd']
    cls = case['cls']
    if cls == 'Conv2dLayer':
        return [RN.conv2d_layer(sd, 'L', xs[0], activation=c.get('activation', 'linear'), up=c.get('up', 1), down=c.get('down', 1),
                                conv_clamp=c.get('conv_clamp'), gain=f.get('gain', 1))]
    if cls == 'SynthesisLayer':
        return [RN.synthesis_layer(sd, 'L', xs[0], xs[1], up=c.get('up', 1), noise_mode=f['noise_mode'], conv_clamp=c.get('conv_clamp'),
                                   gain=f.get('gain', 1), fused_modconv=f['fused_modconv'])]
    if cls == 'ToRGBLayerFull':
        y, parsing = RN.torgb_full(sd, 'L', xs[0], xs[1], conv_clamp=c.get('conv_clamp'), fused_modconv=f['fused_modconv'])
        return [y] + ([parsing] if parsing is not None else [])
    if cls == 'Spade_Norm_Block':
        return [RN.spade_norm_block(sd, 'L', xs[0], xs[1])]
    if cls == 'Spade_ResBlockV2':
        return [RN.spade_resblock(sd, 'L', xs[0], xs[1])]
    if cls == 'MinibatchStdLayer':
        return [RN.minibatch_std(xs[0], group_size=c['group_size'], num_channels=c['num_channels'])]
    return None


@pytest.mark.parametrize('idx', range(len(PC.CASES)), ids=CASE_IDS)
def test_oracle_layers_match_reference(idx):
    case = PC.CASES[idx]
    g = load_golden('layers_primitives.npz')
    _, sd = _state(case)
    xs = [t.requires_grad_(True) for t in PC.case_inputs(case, idx)]
    outs = _oracle_forward(case, sd, xs)
    if outs is None:
        pytest.skip('no stand-alone oracle function for ' + case['cls'] + ' (covered at model level)')
    probe = 0
    for k, o in enumerate(outs):
        assert rel_err(o, g[f"{case['name']}.y{k}"]) < TOL, (case['name'], k)
        probe = probe + (o * PC.rnd(list(o.shape), 7900 + 10 * idx + k)).sum()
    grads = torch.autograd.grad(probe, xs, allow_unused=True)
    for k, gx in enumerate(grads):
        ref = g[f"{case['name']}.dx{k}"]
        gx = gx if gx is not None else torch.zeros_like(xs[k])
        assert rel_err(gx, ref) < TOL or float(np.abs(ref).max()) == 0.0, (case['name'], 'dx', k)


def test_fixture_covers_every_case():
    g = load_golden('layers_primitives.npz')
    for case in PC.CASES:
        assert case['name'] + '.y0' in g and case['name'] + '.dx0' in g
        for p in case['grads']:
            assert case['name'] + '.g.' + p in g


@pytest.mark.gpu
@pytest.mark.parametrize('idx', range(len(PC.CASES)), ids=CASE_IDS)
def test_hip_layers_match_reference(idx):
    from training import networks
    case = PC.CASES[idx]
    g = load_golden('layers_primitives.npz')
    res = PC.run_case(networks, case, idx, device='cuda')
    keys = [k for k in g if k.startswith(case['name'] + '.')]
    assert len(keys) == len(res)
    for k in keys:
        short = k[len(case['name']) + 1:]
        ref = g[k]
        assert rel_err(res[short], ref) < TOL or float(np.abs(ref).max()) == 0.0, (case['name'], short, rel_err(res[short], ref))


@pytest.mark.gpu
def test_spade_block_batched_convolutions_equal_the_separate_ones():
    """Spade_ResBlockV2 with the three conv_mlp / gamma | beta convolutions of its normalisations batched into two launches (round 4) against
    the same block running them one by one (``networks._SPADE_BATCH = False``): output and every gradient, feature map included."""
    import torch
    from training import networks
    g = torch.Generator().manual_seed(11)
    blk = networks.Spade_ResBlockV2(32, 32, conv_clamp=256, resolution=64, feat_channels=48).cuda()
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn(p.shape, generator=g).to(p.device) * (0.3 if p.ndim > 1 else 0.1))
    x = torch.randn([3, 32, 64, 64], generator=g).cuda().requires_grad_(True)
    feat = torch.randn([3, 48, 64, 64], generator=g).cuda().requires_grad_(True)
    dy = torch.randn([3, 32, 64, 64], generator=g).cuda()
    params = list(blk.parameters())
    def run(batched):
        old, networks._SPADE_BATCH = networks._SPADE_BATCH, batched
        try:
            y = blk(x, feat)
            return (y,) + torch.autograd.grad(y, [x, feat] + params, dy)
        finally:
            networks._SPADE_BATCH = old
    a, b = run(True), run(False)
    rel = lambda u, v: float((u.double() - v.double()).abs().max() / (v.double().abs().max() + 1e-30))
    assert rel(a[0], b[0]) < 2e-6
    for i, (u, v) in enumerate(zip(a[1:], b[1:])):
        assert u.shape == v.shape and rel(u, v) < 2e-5, i
