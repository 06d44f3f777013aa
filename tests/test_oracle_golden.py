"""The oracle (oracle/ref_ops.py) against fixtures produced by the reference itself."""

import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_ops as R

TOL = 2e-6   # fp32 rounding of a different summation order


def t(a, grad=False):
    x = torch.from_numpy(np.asarray(a)).clone()
    return x.requires_grad_(True) if grad else x


def _cases(fname):
    g = load_golden(fname)
    return g, json.loads(str(g['manifest']))


def test_setup_filter():
    g, specs = _cases('ops_setup_filter.npz')
    for i, s in enumerate(specs):
        got = R.setup_filter(**s)
        assert rel_err(got, g[f'f{i}']) < 1e-7, s


@pytest.mark.parametrize('as_conv', [False, True])
@pytest.mark.parametrize('idx', range(18))
def test_upfirdn2d(idx, as_conv, monkeypatch):
    monkeypatch.setattr(R, 'FIR_AS_DEPTHWISE_CONV', as_conv)
    g, cases = _cases('ops_upfirdn2d.npz')
    c = cases[idx]
    n = c['name']
    x = t(g[n + '.x'], True)
    f = t(g[n + '.f']) if c['f'] is not None else None
    y = R.upfirdn2d(x, f, **c['call'])
    assert rel_err(y, g[n + '.y']) < TOL
    dx, = torch.autograd.grad(y, x, t(g[n + '.dy']))
    assert rel_err(dx, g[n + '.dx']) < TOL


def test_upfirdn2d_case_count():
    assert len(_cases('ops_upfirdn2d.npz')[1]) == 18


@pytest.mark.parametrize('idx', range(24))
def test_bias_act(idx):
    g, cases = _cases('ops_bias_act.npz')
    c = cases[idx]
    n = c['name']
    x = t(g[n + '.x'], True)
    b = t(g[n + '.b'], True) if c['bias'] else None
    dy = t(g[n + '.dy'], True)
    y = R.bias_act(x, b, act=c['act'], **c['kw'])
    assert rel_err(y, g[n + '.y']) < TOL
    grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dy, create_graph=True)
    assert rel_err(grads[0], g[n + '.dx']) < TOL
    if b is not None:
        assert rel_err(grads[1], g[n + '.db']) < 1e-5
    gg = torch.autograd.grad(grads[0], [dy, x], t(g[n + '.ddx']), allow_unused=True)
    assert rel_err(gg[0], g[n + '.g_dy']) < TOL
    if gg[1] is not None:
        assert rel_err(gg[1], g[n + '.g_x']) < 1e-5


def test_bias_act_case_count():
    assert len(_cases('ops_bias_act.npz')[1]) == 24


@pytest.mark.parametrize('fast', [True, False])
@pytest.mark.parametrize('idx', range(15))
def test_conv2d_resample(idx, fast):
    g, cases = _cases('ops_conv2d_resample.npz')
    c = cases[idx]
    n = c['name']
    x, w = t(g[n + '.x'], True), t(g[n + '.w'], True)
    f = R.setup_filter(c['f']) if c.get('f') is not None else None
    y = R.conv2d_resample(x, w, f=f, fast=fast, **c['kw'])
    assert rel_err(y, g[n + '.y']) < 1e-5
    dx, dw = torch.autograd.grad(y, [x, w], t(g[n + '.dy']))
    assert rel_err(dx, g[n + '.dx']) < 1e-5
    assert rel_err(dw, g[n + '.dw']) < 1e-5


def test_fma():
    g = load_golden('ops_fma.npz')
    a, b, c = t(g['a'], True), t(g['b'], True), t(g['c'], True)
    y = R.fma(a, b, c)
    assert rel_err(y, g['y']) < 1e-6
    da, db, dc = torch.autograd.grad(y, [a, b, c], t(g['dy']))
    assert rel_err(da, g['da']) < 1e-6 and rel_err(db, g['db']) < 1e-6 and rel_err(dc, g['dc']) < 1e-6


# ----------------------------------------------------------------------------- layers / models

@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('idx', range(5))
def test_modulated_conv2d(idx, fused):
    g, cases = _cases('layers_modconv.npz')
    c = cases[idx]
    n = c['name']
    x, w, s = t(g[n + '.x'], True), t(g[n + '.w'], True), t(g[n + '.s'], True)
    noise = t(g[n + '.noise']) if c['noise'] is not None else None
    f = R.setup_filter(c['f']) if c.get('f') is not None else None
    y = R.modulated_conv2d(x, w, s, noise=noise, resample_filter=f, fused_modconv=fused, **c['kw'])
    tag = n + ('.fused' if fused else '.plain')
    assert rel_err(y, g[tag + '.y']) < 1e-5
    dx, dw, ds = torch.autograd.grad(y, [x, w, s], t(g[n + '.dy']))
    assert rel_err(dx, g[tag + '.dx']) < 1e-5 and rel_err(dw, g[tag + '.dw']) < 1e-5 and rel_err(ds, g[tag + '.ds']) < 1e-5


def _check_summary(g, key, tensor, tol):
    from oracle import param_fill as PF
    s = PF.summarize(tensor)
    assert rel_err(s['sample'], g[key + '.sample']) < tol, key
    m, mg = s['moments'], g[key + '.moments']
    assert abs(m[1] - mg[1]) <= tol * abs(mg[1]) + 1e-12, key       # sum |x|
    assert abs(m[2] - mg[2]) <= 2 * tol * abs(mg[2]) + 1e-12, key   # sum x^2


def _product_state_dict(kind):
    """Parameter names/shapes come from this repo's module constructors (CPU), values from param_fill."""
    from oracle import param_fill as PF
    from training import networks
    cls, kw = (networks.GeneratorFull, PF.G_KWARGS) if kind == 'G' else (networks.Discriminator, PF.D_KWARGS)
    m = PF.fill_module(cls(**kw))
    sd = {k: v.detach().clone() for k, v in list(m.named_parameters()) + list(m.named_buffers())}
    for k, v in m.named_parameters():
        sd[k].requires_grad_(True)
    return sd


def test_generator_full_oracle():
    from oracle import param_fill as PF, ref_networks as RN
    g = load_golden('models_fullbody.npz')
    sd = _product_state_dict('G')
    inp = PF.make_inputs(n=2, seed=0)
    args = (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
            inp['denorm_upper_mask'], inp['denorm_lower_mask'])
    img, fin, par = RN.generator_full(sd, *args, **_g_cfg())
    _check_summary(g, 'G.img', img, 1e-4)
    _check_summary(g, 'G.pred_parsing', par, 1e-4)
    _check_summary(g, 'G.finetune_img', fin, 1e-4)
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    assert abs(probe.item() - float(g['G.probe'][0])) < 1e-4 * abs(float(g['G.probe'][0]))
    from oracle.make_golden_models import GRAD_KEYS_G
    grads = torch.autograd.grad(probe, [sd[k] for k in GRAD_KEYS_G])
    for k, gr in zip(GRAD_KEYS_G, grads):
        _check_summary(g, 'G.grad.' + k, gr, 1e-3)
    with torch.no_grad():
        img_e, fin_e, _ = RN.generator_full(sd, *args, fused_modconv=True, **_g_cfg())
    _check_summary(g, 'G.eval.img', img_e, 1e-4)
    _check_summary(g, 'G.eval.finetune_img', fin_e, 1e-4)


def _g_cfg():
    return dict(img_resolution=256, conv_clamp=256, mapping_layers=1, noise_mode='const')


def test_discriminator_oracle():
    from oracle import param_fill as PF, ref_networks as RN
    from oracle.make_golden_models import GRAD_KEYS_D
    g = load_golden('models_fullbody.npz')
    sd = _product_state_dict('D')
    c = torch.tanh(PF.make_inputs(n=4, seed=1)['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
    x = PF.make_inputs(n=4, seed=1)['real_img'].requires_grad_(True)
    logits = RN.discriminator(sd, x, c)
    assert rel_err(logits, g['D.logits']) < 1e-4
    gx, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    _check_summary(g, 'D.r1_grads', gx, 1e-4)
    pen = gx.square().sum([1, 2, 3])
    assert rel_err(pen, g['D.r1_penalty']) < 1e-4
    loss = torch.nn.functional.softplus(-logits).mean() + 5.0 * pen.mean()
    grads = torch.autograd.grad(loss, [sd[k] for k in GRAD_KEYS_D])
    for k, gr in zip(GRAD_KEYS_D, grads):
        _check_summary(g, 'D.grad.' + k, gr, 1e-3)


def _v18_state_dict():
    from oracle import param_fill as PF
    from training import networks
    m = PF.fill_module(networks.GeneratorV18(**PF.G_KWARGS))
    return {k: v.detach().clone() for k, v in list(m.named_parameters()) + list(m.named_buffers())}


def test_generator_v18_oracle():
    from oracle import param_fill as PF, ref_networks as RN
    g = load_golden('models_fullbody.npz')
    sd = _v18_state_dict()
    inp = PF.make_inputs(n=2, seed=0)
    c60 = PF.make_inputs(n=2, seed=5)['style_input'].repeat(1, 2, 1, 1)[:, :60]
    with torch.no_grad():
        outs = RN.generator_v18(sd, inp['gen_z'], c60, inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                                inp['denorm_upper_mask'], inp['denorm_lower_mask'], fused_modconv=True, **_g_cfg())
    for name, tns in zip(['img', 'finetune_img', 'upper_mask', 'lower_mask'], outs):
        _check_summary(g, 'G18.' + name, tns, 1e-4)
