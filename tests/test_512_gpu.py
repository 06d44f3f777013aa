"""BASELINE config 4 (512x320 generator inference, batch 8): the operators at 512 / 513-pixel planes against the oracle
(on slices the oracle finishes in seconds) and through size-independent properties at the full batch-8 sizes; the
inference path (eval mode: per-sample weights, grouped convolution) of the resolution-generalised generator against the
oracle's restatement of that generalisation (parity UNPINNED above 256: the reference ships no 512 class);
and the benchmark's inference mode."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_err
from oracle import param_fill as PF

pytestmark = pytest.mark.gpu

F4 = [1, 3, 3, 1]

UPF_512 = [   # the resampling calls of a 512 generator: blur behind a transposed convolution, RGB up, skip down, plain blur
    ('blur_up_tail', [8, 32, 513, 513], dict(padding=[1, 1, 1, 1], gain=4)),
    ('blur_pad2',    [8, 32, 512, 512], dict(padding=[2, 2, 2, 2])),
    ('rgb_up2',      [8, 3, 256, 256],  dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    ('down2',        [8, 32, 512, 512], dict(down=2, padding=[1, 1, 1, 1])),
]


@pytest.mark.parametrize('name,shape,kw', UPF_512, ids=[u[0] for u in UPF_512])
def test_upfirdn2d_512_planes(name, shape, kw):
    from oracle import ref_ops as R
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(len(name))
    f = upfirdn2d.setup_filter(F4)
    x = torch.randn(shape, generator=g)
    xg = x.cuda().requires_grad_(True)
    y = upfirdn2d.upfirdn2d(xg, f.cuda(), **kw)
    # oracle on two planes of the batch (first and last: both ends of the grid)
    sl = x[[0, -1]][:, [0, -1]].clone().requires_grad_(True)
    yr = R.upfirdn2d(sl, f, **kw)
    assert rel_err(y[[0, -1]][:, [0, -1]], yr) < 1e-5
    dy = torch.randn(y.shape, generator=g)
    dx, = torch.autograd.grad(y, xg, dy.cuda())
    dxr, = torch.autograd.grad(yr, sl, dy[[0, -1]][:, [0, -1]])
    assert rel_err(dx[[0, -1]][:, [0, -1]], dxr) < 1e-5
    # adjoint identity over the whole batch
    lhs, rhs = float((dy.cuda().double() * y.double()).sum()), float((dx.double() * xg.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)


CONV_512 = [
    ('3x3_512',       [8, 32, 512, 512], [32, 32, 3, 3], dict(padding=1)),
    ('1x1_torgb_512', [8, 32, 512, 512], [3, 32, 1, 1], dict()),
    ('up2_256to512',  [8, 64, 256, 256], [32, 64, 3, 3], dict(up=2, padding=1, flip_weight=False)),
    ('grouped_3x3',   [1, 8 * 32, 512, 512], [8 * 32, 32, 3, 3], dict(padding=1, groups=8)),      # eval mode: batch as groups
]


@pytest.mark.parametrize('name,xs,ws,kw', CONV_512, ids=[c[0] for c in CONV_512])
def test_conv2d_resample_512_planes(name, xs, ws, kw):
    from oracle import ref_ops as R
    from torch_utils.ops import conv2d_resample, upfirdn2d
    g = torch.Generator().manual_seed(len(name))
    f = upfirdn2d.setup_filter(F4)
    x = torch.randn(xs, generator=g)
    w = torch.randn(ws, generator=g) / np.sqrt(ws[1] * ws[2] * ws[3])
    y = conv2d_resample.conv2d_resample(x.cuda(), w.cuda(), f=f.cuda(), **kw)
    groups = kw.get('groups', 1)
    if groups == 1:         # one sample through the oracle
        yr = R.conv2d_resample(x[-1:], w, f=f, **kw)
        assert rel_err(y[-1:], yr) < 1e-5
    else:                   # one group through the oracle
        ci, co = xs[1] // groups, ws[0] // groups
        yr = R.conv2d_resample(x[:, -ci:], w[-co:], f=f, **{k: v for k, v in kw.items() if k != 'groups'})
        assert rel_err(y[:, -co:], yr) < 1e-5
    with torch.no_grad():   # linearity in x at the full size
        x2 = torch.randn(xs, generator=g).cuda()
        a = conv2d_resample.conv2d_resample(2 * x.cuda() - 3 * x2, w.cuda(), f=f.cuda(), **kw)
        b = 2 * y - 3 * conv2d_resample.conv2d_resample(x2, w.cuda(), f=f.cuda(), **kw)
        assert float((a - b).abs().max() / a.abs().max()) < 1e-4


G512 = dict(z_dim=0, c_dim=512, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=1),
            synthesis_kwargs=dict(channel_base=2048, channel_max=512, conv_clamp=256))


def _inputs(n):
    inp = PF.make_inputs(n=n, seed=0, res=512)
    return inp, (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                 inp['denorm_upper_mask'], inp['denorm_lower_mask'])


def test_generator_full_512_structure_and_inference_vs_oracle():
    """The 512 model: one more pyramid level everywhere (7 pose stages -> 4x4, five retained-image features for the merges
    at 32..512, SPADE stage at 256, texture block at 512); eval-mode output against the oracle's restatement."""
    from oracle import ref_networks as RN
    from training import networks
    G = PF.fill_module(networks.GeneratorFull(**G512)).eval().requires_grad_(False)
    names = [n for n, _ in G.synthesis.named_children()]
    assert names == ['b4', 'b8', 'b16', 'b32', 'b64', 'b128', 'b256', 'b512', 'spade_b256_1', 'spade_b256_2', 'spade_b256_3', 'texture_b512', 'spade_encoder']
    assert len(G.const_encoding.model) == 8 and len(G.style_encoding.feat_enc) == 5 and G.num_ws == 16
    sd = {k: v.detach() for k, v in list(G.named_parameters()) + list(G.named_buffers())}
    inp, args = _inputs(1)
    with torch.no_grad():
        want = RN.generator_full(sd, *args, img_resolution=512, conv_clamp=256, mapping_layers=1, noise_mode='const', fused_modconv=True)
        got = G.cuda()(*[a.cuda() for a in args], noise_mode='const')
    for name, a, b in zip(['img', 'finetune_img', 'pred_parsing'], got, want):
        assert a.shape == b.shape == (1, b.shape[1], 512, 512)
        assert rel_err(a, b) < 1e-4, name


def test_generator_full_512_training_mode_vs_oracle():
    """Same model in training mode (shared-weight modulated convolutions, fused SPADE / demodulation kernels), gradients of
    a scalar probe into a parameter of every part."""
    from oracle import ref_networks as RN
    from training import networks
    G = PF.fill_module(networks.GeneratorFull(**G512)).train().requires_grad_(True)
    params = dict(G.named_parameters())
    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in list(G.named_parameters()) + list(G.named_buffers())}
    inp, args = _inputs(1)
    keys = ['synthesis.b512.conv0.weight', 'synthesis.spade_b256_2.spade0.conv_gamma.weight', 'synthesis.texture_b512.conv1.weight',
            'const_encoding.model.7.weight', 'style_encoding.feat_enc.4.weight', 'synthesis.b32.merge_conv.weight']
    img, fin, par = RN.generator_full(sd, *args, img_resolution=512, conv_clamp=256, mapping_layers=1, noise_mode='const')
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    want = torch.autograd.grad(probe, [sd[k] for k in keys])
    G = G.cuda()
    gi, gf, gp = G(*[a.cuda() for a in args], noise_mode='const')
    assert rel_err(gi, img) < 1e-4 and rel_err(gf, fin) < 1e-4 and rel_err(gp, par) < 1e-4
    ((gi * inp['real_img'].cuda()).mean() + gf.square().mean() + 0.1 * gp.abs().mean()).backward()
    got = dict(G.named_parameters())
    for k, w in zip(keys, want):
        assert rel_err(got[k].grad, w) < 1e-3, k


def test_generator_v18_256_unchanged_by_the_generalisation():
    """At 256 the generalised classes ARE the reference's: same module names as the pinned fixture's model."""
    from training import networks
    G = networks.GeneratorV18(**PF.G_KWARGS)
    names = [n for n, _ in G.synthesis.named_children()]
    assert names == ['b4', 'b8', 'b16', 'b32', 'b64', 'b128', 'b256', 'spade_b128_1', 'spade_b128_2', 'spade_b128_3', 'texture_b256', 'spade_encoder']
    assert len(G.const_encoding.model) == 7 and len(G.style_encoding.feat_enc) == 4


@pytest.mark.timeout(900)
@pytest.mark.parametrize('res', [256, 512])
def test_bench_inference_mode(res):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'infer', '--res', str(res), '--steps', '2', '--warmup', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['value'] > 0 and out['config']['global_batch'] == 8 and 'config 4' in out['config']['workload']
    assert out['roofline']['bound'] == 'hbm' and 0 < out['roofline']['frac'] < 1
    assert ('UNPINNED' in out['config']['workload']) == (res == 512)
