"""Generator / discriminator on the HIP path against the reference's golden vectors (and the oracle)."""

import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import param_fill as PF
from oracle.make_golden_models import GRAD_KEYS_D, GRAD_KEYS_G

pytestmark = pytest.mark.gpu

# Parity bar of BASELINE.json: 1e-3 relative in fp32 on fixed inputs.  Forward tensors are checked
# 10x tighter; parameter gradients (sums over up to 10^5 pixels) at the stated 1e-3.
TOL_FWD = 1e-4
TOL_GRAD = 1e-3


def _summary_ok(g, key, tensor, tol):
    s = PF.summarize(tensor)
    e = rel_err(s['sample'], g[key + '.sample'])
    m, mg = s['moments'], g[key + '.moments']
    assert e < tol, (key, e)
    assert abs(m[1] - mg[1]) <= tol * abs(mg[1]) + 1e-12, key
    assert abs(m[2] - mg[2]) <= 2 * tol * abs(mg[2]) + 1e-12, key


def _grad_ok(g, g64, key, tensor, tol):
    """A parameter gradient meets the bar against the reference's fp32 evaluation (``models_fullbody.npz``) OR against the
    reference's own code run in fp64 (``models_fullbody_f64.npz``, oracle/make_golden_models_f64.py).  Some gradients of the
    closed-form-filled generator are cancellation residue (the demodulated style path): on ``synthesis.b64.conv0.affine.weight`` the
    reference's fp32 result is 1.25e-3 from its fp64 result, and so are fp32 MFMA (1.1e-3) and split-bf16 (1.5e-3), while the
    three-product fp16 arithmetic lands 1.5e-4 from the fp64 value (tools/diag_modes.py, measured on MI355X) -- 1e-3 against ONE fp32
    rounding of such a quantity would measure luck, 1e-3 against either evaluation measures parity with the reference's algorithm."""
    s = PF.summarize(tensor)
    e32 = rel_err(s['sample'], g[key + '.sample'])
    e64 = rel_err(s['sample'].astype(np.float64), g64[key + '.sample'])
    assert min(e32, e64) < tol, (key, e32, e64)
    if e32 < tol:
        m, mg = s['moments'], g[key + '.moments']
        assert abs(m[1] - mg[1]) <= tol * abs(mg[1]) + 1e-12 and abs(m[2] - mg[2]) <= 2 * tol * abs(mg[2]) + 1e-12, key
    return e32, e64


def _cuda(inp):
    return {k: v.cuda() for k, v in inp.items()}


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('idx', range(5))
def test_modulated_conv2d_golden(idx, fused):
    from training import networks
    from torch_utils.ops import upfirdn2d
    g = load_golden('layers_modconv.npz')
    c = json.loads(str(g['manifest']))[idx]
    n = c['name']
    dev = lambda a, grad=False: torch.from_numpy(np.asarray(a)).cuda().requires_grad_(grad)
    x, w, s = dev(g[n + '.x'], True), dev(g[n + '.w'], True), dev(g[n + '.s'], True)
    noise = dev(g[n + '.noise']) if c['noise'] is not None else None
    f = upfirdn2d.setup_filter(c['f']).cuda() if c.get('f') is not None else None
    y = networks.modulated_conv2d(x=x, weight=w, styles=s, noise=noise, resample_filter=f, fused_modconv=fused, **c['kw'])
    tag = n + ('.fused' if fused else '.plain')
    assert rel_err(y, g[tag + '.y']) < 1e-5
    dx, dw, ds = torch.autograd.grad(y, [x, w, s], dev(g[n + '.dy']))
    assert rel_err(dx, g[tag + '.dx']) < 1e-5 and rel_err(dw, g[tag + '.dw']) < 2e-5 and rel_err(ds, g[tag + '.ds']) < 2e-5


def test_generator_full_golden(arith):
    from training import networks
    g = load_golden('models_fullbody.npz')
    g64 = load_golden('models_fullbody_f64.npz')
    G = PF.fill_module(networks.GeneratorFull(**PF.G_KWARGS)).cuda().train().requires_grad_(True)
    inp = _cuda(PF.make_inputs(n=2, seed=0))
    args = (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
            inp['denorm_upper_mask'], inp['denorm_lower_mask'])
    img, fin, par = G(*args, noise_mode='const')
    _summary_ok(g, 'G.img', img, TOL_FWD)
    _summary_ok(g, 'G.pred_parsing', par, TOL_FWD)
    _summary_ok(g, 'G.finetune_img', fin, TOL_FWD)
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    assert abs(probe.item() - float(g['G.probe'][0])) < TOL_FWD * abs(float(g['G.probe'][0]))
    probe.backward()
    sd = dict(G.named_parameters())
    errs = [_grad_ok(g, g64, 'G.grad.' + k, sd[k].grad, TOL_GRAD) for k in GRAD_KEYS_G]
    # at most one key needs the fp64 evaluation, under either arithmetic (synthesis.b64.conv0.affine.weight: split-bf16 1.5e-3 and the
    # three-product default 1.3e-3 from the reference's fp32 value, which is itself 1.25e-3 from the reference's fp64 value)
    assert sum(e32 < TOL_GRAD for e32, _ in errs) >= len(errs) - 1, (arith, errs)
    G.eval()
    with torch.no_grad():
        img_e, fin_e, _ = G(*args, noise_mode='const')
    _summary_ok(g, 'G.eval.img', img_e, TOL_FWD)
    _summary_ok(g, 'G.eval.finetune_img', fin_e, TOL_FWD)


def test_discriminator_golden_with_r1():
    from training import networks
    from torch_utils.ops import conv2d_gradfix
    g = load_golden('models_fullbody.npz')
    D = PF.fill_module(networks.Discriminator(**PF.D_KWARGS)).cuda().train().requires_grad_(True)
    c = torch.tanh(PF.make_inputs(n=4, seed=1)['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512]).cuda()
    x = PF.make_inputs(n=4, seed=1)['real_img'].cuda().requires_grad_(True)
    logits = D(x, c)
    assert rel_err(logits, g['D.logits']) < TOL_FWD
    with conv2d_gradfix.no_weight_gradients():
        gx, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    _summary_ok(g, 'D.r1_grads', gx, TOL_FWD)
    pen = gx.square().sum([1, 2, 3])
    assert rel_err(pen, g['D.r1_penalty']) < TOL_FWD
    loss = torch.nn.functional.softplus(-logits).mean() + 5.0 * pen.mean()
    loss.backward()
    sd = dict(D.named_parameters())
    for k in GRAD_KEYS_D:
        _summary_ok(g, 'D.grad.' + k, sd[k].grad, TOL_GRAD)


def test_generator_random_noise_and_state_dict_names():
    """noise_mode='random' runs (not comparable across devices), and parameter names match the golden run's."""
    from training import networks
    G = networks.GeneratorFull(**PF.G_KWARGS).cuda()
    inp = _cuda(PF.make_inputs(n=2, seed=3))
    img, fin, par = G(inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'],
                      inp['denorm_lower_input'], inp['denorm_upper_mask'], inp['denorm_lower_mask'])
    assert img.shape == (2, 3, 256, 256) and fin.shape == (2, 3, 256, 256) and par.shape == (2, 6, 256, 256)
    assert torch.isfinite(img).all() and torch.isfinite(fin).all()
    g = load_golden('models_fullbody.npz')
    assert len(dict(G.named_parameters())) == len(g['G.gradnorms'])


def test_generator_v18_inference_golden():
    """The released 256 model's class (test.py:120-128): eval mode, fused modulated (grouped) convolutions, sigmoid heads."""
    from training import networks
    g = load_golden('models_fullbody.npz')
    G = PF.fill_module(networks.GeneratorV18(**PF.G_KWARGS)).cuda().eval().requires_grad_(False)
    inp = _cuda(PF.make_inputs(n=2, seed=0))
    c60 = PF.make_inputs(n=2, seed=5)['style_input'].repeat(1, 2, 1, 1)[:, :60].cuda()
    with torch.no_grad():
        outs = G(inp['gen_z'], c60, inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                 inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    for name, t in zip(['img', 'finetune_img', 'upper_mask', 'lower_mask'], outs):
        _summary_ok(g, 'G18.' + name, t, TOL_FWD)


def test_vgg19_perceptual_term_matches_the_cpu_restatement():
    """Row f2: VGG19_Feature / VGGLoss on the HIP convolution against oracle/ref_vgg.py with the same (seeded random)
    weights: the five feature maps, the loss value and its gradient with respect to the image."""
    from oracle import ref_vgg
    from training.loss_wo_flow_fullbody import VGGLoss
    crit = VGGLoss(torch.device('cuda'), random_init=True)
    state = {}
    for name, buf in crit.vgg.named_buffers():
        _, idx, kind = name.split('_')
        state[f'features.{idx}.{kind}'] = buf.detach().cpu()
    assert len(state) == 26 and state['features.28.weight'].shape == (512, 512, 3, 3)      # 13 convolutions up to relu5_1
    g = torch.Generator().manual_seed(4)
    x = torch.rand([2, 3, 64, 64], generator=g) * 2 - 1
    y = torch.rand([2, 3, 64, 64], generator=g) * 2 - 1
    xr = x.clone().requires_grad_(True)
    ref = ref_vgg.vgg_loss(xr, y, state)
    gref, = torch.autograd.grad(ref, xr)
    feats = crit.vgg(x.cuda())
    for a, b in zip(feats, ref_vgg.features(x, state)):
        assert a.shape == b.shape and rel_err(a, b) < 1e-5
    xc = x.cuda().requires_grad_(True)
    ours = crit(xc, y.cuda())
    gours, = torch.autograd.grad(ours, xc)
    assert abs(float(ours) - float(ref)) < 1e-5 * abs(float(ref))
    # d|a - b| is a sign: elements whose difference is rounding noise may flip it, so the gradient of the loss is compared
    # in direction and norm, and the backward path itself through a smooth (linear) functional of the features
    cos = float((gours.cpu() * gref).sum() / (gours.norm().cpu() * gref.norm()))
    assert cos > 0.9999 and abs(float(gours.norm()) / float(gref.norm()) - 1) < 1e-3
    probes = [torch.randn(list(f.shape), generator=g) for f in feats]
    xr2 = x.clone().requires_grad_(True)
    lin_ref = sum((f * r).sum() for f, r in zip(ref_vgg.features(xr2, state), probes))
    g_ref, = torch.autograd.grad(lin_ref, xr2)
    xc2 = x.cuda().requires_grad_(True)
    lin = sum((f * r.cuda()).sum() for f, r in zip(crit.vgg(xc2), probes))
    g_ours, = torch.autograd.grad(lin, xc2)
    assert rel_err(g_ours, g_ref) < 1e-3
