"""TEST INFRASTRUCTURE (oracle side): the layer-level parity cases of SURVEY.md 8c -- Conv2dLayer, SynthesisLayer,
ToRGBLayerFull, Spade_Norm_Block, Spade_ResBlockV2, MinibatchStdLayer, DiscriminatorBlock with small channels.

``run_case(nets, case, device)`` builds the layer from any module that has the reference's class names and
constructor signatures (the reference's ``training.networks`` when ``oracle/make_golden.py`` writes the fixture,
this repository's overlay in the GPU tests), fills it with the closed-form weights of ``param_fill``, runs forward
and backward on seeded inputs and returns every tensor a parity test compares."""

import numpy as np
import torch

from oracle import param_fill as PF


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


# name, class, constructor kwargs, input shapes (in call order), forward kwargs, parameters whose gradients are kept
CASES = [
    dict(name='conv_lrelu', cls='Conv2dLayer', ctor=dict(in_channels=5, out_channels=7, kernel_size=3, activation='lrelu'),
         inputs=[[2, 5, 16, 16]], fwd=dict(), grads=['weight', 'bias']),
    dict(name='conv_down_clamp', cls='Conv2dLayer', ctor=dict(in_channels=6, out_channels=4, kernel_size=3, activation='lrelu', down=2, conv_clamp=0.6),
         inputs=[[2, 6, 16, 16]], fwd=dict(gain=float(np.sqrt(0.5))), grads=['weight', 'bias']),
    dict(name='conv_skip_1x1_down', cls='Conv2dLayer', ctor=dict(in_channels=6, out_channels=8, kernel_size=1, bias=False, down=2),
         inputs=[[2, 6, 16, 16]], fwd=dict(gain=float(np.sqrt(0.5))), grads=['weight']),
    dict(name='conv_up', cls='Conv2dLayer', ctor=dict(in_channels=4, out_channels=6, kernel_size=3, activation='relu', up=2),
         inputs=[[2, 4, 8, 8]], fwd=dict(), grads=['weight', 'bias']),
    dict(name='conv_7x7_stem', cls='Conv2dLayer', ctor=dict(in_channels=3, out_channels=8, kernel_size=7, activation='relu'),
         inputs=[[2, 3, 20, 20]], fwd=dict(), grads=['weight', 'bias']),
    dict(name='synth_const', cls='SynthesisLayer', ctor=dict(in_channels=6, out_channels=8, w_dim=12, resolution=16, conv_clamp=1.5),
         inputs=[[2, 6, 16, 16], [2, 12]], fwd=dict(noise_mode='const', fused_modconv=False), grads=['weight', 'bias', 'affine.weight', 'noise_strength']),
    dict(name='synth_up_none', cls='SynthesisLayer', ctor=dict(in_channels=8, out_channels=5, w_dim=12, resolution=16, up=2),
         inputs=[[2, 8, 8, 8], [2, 12]], fwd=dict(noise_mode='none', fused_modconv=False, gain=float(np.sqrt(0.5))), grads=['weight', 'affine.bias']),
    dict(name='synth_eval_fused', cls='SynthesisLayer', ctor=dict(in_channels=6, out_channels=6, w_dim=12, resolution=8),
         inputs=[[3, 6, 8, 8], [3, 12]], fwd=dict(noise_mode='const', fused_modconv=True), grads=['weight']),
    dict(name='torgb_plain', cls='ToRGBLayerFull', ctor=dict(in_channels=8, out_channels=3, w_dim=12, conv_clamp=0.8),
         inputs=[[2, 8, 16, 16], [2, 12]], fwd=dict(fused_modconv=False), grads=['weight', 'bias', 'affine.weight']),
    dict(name='torgb_last_style', cls='ToRGBLayerFull', ctor=dict(in_channels=8, out_channels=3, w_dim=12, conv_clamp=256, is_last=True, is_style=True),
         inputs=[[2, 8, 16, 16], [2, 12]], fwd=dict(fused_modconv=False), grads=['weight', 'm_weight1', 'm_bias1']),
    dict(name='spade_norm', cls='Spade_Norm_Block', ctor=dict(in_channels=10, norm_channels=6),
         inputs=[[2, 6, 16, 16], [2, 10, 16, 16]], fwd=dict(), grads=['conv_mlp.weight', 'conv_gamma.weight', 'conv_beta.weight']),
    dict(name='spade_resblock', cls='Spade_ResBlockV2', ctor=dict(in_channels=8, out_channels=12, conv_clamp=256, resolution=64),
         inputs=[[2, 8, 16, 16], [2, 128, 16, 16]], fwd=dict(), grads=['conv.weight', 'skip.weight', 'spade1.conv_beta.weight', 'conv1.weight']),
    dict(name='mbstd', cls='MinibatchStdLayer', ctor=dict(group_size=4, num_channels=2),
         inputs=[[8, 6, 4, 4]], fwd=dict(), grads=[]),
    dict(name='dblock_resnet', cls='DiscriminatorBlock', ctor=dict(in_channels=6, tmp_channels=6, out_channels=10, resolution=16, img_channels=3,
                                                                  first_layer_idx=2, architecture='resnet', conv_clamp=256),
         inputs=[[4, 6, 16, 16]], fwd=dict(), grads=['conv0.weight', 'conv1.bias', 'skip.weight'], dblock=True),
    dict(name='dblock_first', cls='DiscriminatorBlock', ctor=dict(in_channels=0, tmp_channels=6, out_channels=8, resolution=16, img_channels=3,
                                                                 first_layer_idx=0, architecture='resnet', conv_clamp=256),
         inputs=[[4, 3, 16, 16]], fwd=dict(), grads=['fromrgb.weight', 'conv1.weight'], dblock=True, first=True),
]


def case_inputs(case, idx):
    return [rnd(shape, 7000 + 10 * idx + j, 0.8 if j == 0 else 0.5) for j, shape in enumerate(case['inputs'])]


def run_case(nets, case, idx, device='cpu', train=True):
    """-> dict of CPU tensors: outputs 'y0', 'y1', ..., input gradients 'dx0', ..., parameter gradients 'g.<name>'."""
    layer = PF.fill_module(getattr(nets, case['cls'])(**case['ctor'])).to(device)
    layer.train(train and case['name'] != 'synth_eval_fused')
    layer.requires_grad_(True)
    xs = [t.to(device).requires_grad_(True) for t in case_inputs(case, idx)]
    if case.get('dblock'):
        # forward(x, img, force_fp32): the first block consumes the image, the others a feature map
        out = layer(None, xs[0], True) if case.get('first') else layer(xs[0], None, True)
        out = out[0]
    else:
        out = layer(*xs, **case['fwd'])
    outs = [o for o in (out if isinstance(out, (tuple, list)) else [out]) if o is not None]
    probe = 0
    for k, o in enumerate(outs):
        probe = probe + (o * rnd(list(o.shape), 7900 + 10 * idx + k).to(device)).sum()
    params = dict(layer.named_parameters())
    wanted = [params[g] for g in case['grads']]
    grads = torch.autograd.grad(probe, xs + wanted, allow_unused=True)
    res = {}
    for k, o in enumerate(outs):
        res[f'y{k}'] = o.detach().cpu()
    for k, g in enumerate(grads[:len(xs)]):
        res[f'dx{k}'] = (g if g is not None else torch.zeros_like(xs[k])).detach().cpu()
    for name, g in zip(case['grads'], grads[len(xs):]):
        res['g.' + name] = g.detach().cpu()
    return res
