"""BASELINE config 2 at its real layer widths (``cfg=fashion``: channel_base 16384) against vectors written by the
reference's own GeneratorFull / Discriminator (oracle/make_golden_fullwidth.py).  These are the shapes the benchmark runs:
the 128x128 / 64x256 tiles, the row-reuse kernel, split-K on the 4..16-pixel layers, the 3x3 / strided / 1x1 weight-gradient
kernels all meet the reference here, not only adjoint identities.

Tolerances (north_star: 1e-3 relative fp32): forward tensors 1e-4, gradients 1e-3, every parameter's gradient norm 1e-3;
the fp16 discriminator (blocks b256..b32 store and compute in fp16 in the reference, fp16 storage with fp32 accumulation
here) 2e-2 on logits and gradient norms -- the spread of two fp16 evaluation orders, not a precision claim."""

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import param_fill as PF
from oracle import make_golden_fullwidth as FW

TOL_FWD, TOL_GRAD, TOL_FP16 = 1e-4, 1e-3, 2e-2


def _summary_ok(g, key, tensor, tol, samples=FW.SAMPLES):
    s = PF.summarize(tensor, samples=samples)
    e = rel_err(s['sample'], g[key + '.sample'])
    m, mg = s['moments'], g[key + '.moments']
    assert e < tol, (key, e)
    assert abs(m[1] - mg[1]) <= tol * abs(mg[1]) + 1e-12, (key, 'sum |x|')
    assert abs(m[2] - mg[2]) <= 2 * tol * abs(mg[2]) + 1e-12, (key, 'sum x^2')


def _gradnorms_ok(g, key, named_grads, tol, floor=2e-6):
    """Every parameter's gradient norm within ``tol`` relative, plus an absolute slack of ``floor`` x the largest norm:
    the trunk of the style encoder sits behind instance norms, its gradients (1e-6 .. 1e-13 against 0.23 for the largest)
    are cancellation residue and differ by several per cent between two fp32 evaluation orders of the reference's own
    arithmetic (measured: this oracle against the reference, both fp32 on the CPU)."""
    ref = g[key]
    got = np.array([float(v.float().norm()) if v is not None else -1.0 for _, v in sorted(named_grads.items())])
    assert got.shape == ref.shape
    assert ((got < 0) == (ref < 0)).all(), 'different sets of parameters without a gradient'
    scale = np.abs(ref).max()
    bad = [(k, a, b) for (k, _), a, b in zip(sorted(named_grads.items()), got, ref) if abs(a - b) > tol * abs(b) + floor * scale]
    assert not bad, bad[:8]


def _g_args(inp):
    return (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
            inp['denorm_upper_mask'], inp['denorm_lower_mask'])


def _d_inputs():
    c = torch.tanh(PF.make_inputs(n=4, seed=1)['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
    return PF.make_inputs(n=4, seed=1)['real_img'], c


# ---- the oracle restatement itself, on the CPU --------------------------------------------------------------------------

def _state(cls, kw):
    m = PF.fill_module(cls(**kw))
    params = dict(m.named_parameters())
    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in list(m.named_parameters()) + list(m.named_buffers())}
    return sd, sorted(params)


@pytest.mark.timeout(600)
def test_oracle_generator_at_full_width():
    from oracle import ref_networks as RN
    from training import networks
    g = load_golden('models_fullwidth.npz')
    sd, pnames = _state(networks.GeneratorFull, FW.G_KWARGS)
    inp = PF.make_inputs(n=2, seed=0)
    img, fin, par = RN.generator_full(sd, *_g_args(inp), img_resolution=256, conv_clamp=256, mapping_layers=1, noise_mode='const')
    for key, t in [('G.img', img), ('G.finetune_img', fin), ('G.pred_parsing', par)]:
        _summary_ok(g, key, t, TOL_FWD)
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    assert abs(probe.item() - float(g['G.probe'][0])) < TOL_FWD * abs(float(g['G.probe'][0]))
    grads = dict(zip(pnames, torch.autograd.grad(probe, [sd[k] for k in pnames], allow_unused=True)))
    for k in FW.GRAD_KEYS_G:
        _summary_ok(g, 'G.grad.' + k, grads[k], TOL_GRAD)
    _gradnorms_ok(g, 'G.gradnorms', grads, TOL_GRAD)


@pytest.mark.timeout(600)
def test_oracle_discriminator_at_full_width():
    from oracle import ref_networks as RN
    from training import networks
    g = load_golden('models_fullwidth.npz')
    sd, pnames = _state(networks.Discriminator, FW.D_KWARGS)
    x, c = _d_inputs()
    x.requires_grad_(True)
    logits = RN.discriminator(sd, x, c)
    assert rel_err(logits, g['D.logits']) < TOL_FWD
    gx, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    _summary_ok(g, 'D.r1_grads', gx, TOL_FWD)
    pen = gx.square().sum([1, 2, 3])
    # sums of squares of 1e-11-sized gradients: 1e-4 on the host that wrote the fixture, 6e-4 on another CPU model (different
    # convolution blocking in torch's CPU back end), so the bound is the latter's
    assert rel_err(pen, g['D.r1_penalty']) < 2e-3
    loss = torch.nn.functional.softplus(-logits).mean() + 5.0 * pen.mean()
    grads = dict(zip(pnames, torch.autograd.grad(loss, [sd[k] for k in pnames], allow_unused=True)))
    # (the double backward through the R1 term is as sensitive to the host's convolution back end: 1e-3 where the fixture was
    # written, 1.3e-3 on b4.conv.weight on another CPU model)
    for k in FW.GRAD_KEYS_D:
        _summary_ok(g, 'D.grad.' + k, grads[k], 3 * TOL_GRAD)
    _gradnorms_ok(g, 'D.gradnorms', grads, 3 * TOL_GRAD)


def _r1_check(g, logits, gx, pen, grads, tol_fwd, tol_grad, sample_slack=2):
    """Dreg phase against models_fullwidth_r1.npz: every parameter gradient here is a SECOND derivative (the loss is the R1
    term alone), |d logit / d img| ~ 1e-4 and the penalty ~ 7e-5: nothing degenerate."""
    from oracle import make_golden_fullwidth_r1 as R1
    assert float(np.abs(g['Dr1.r1_grads.sample']).max()) > 1e-5 and float(g['Dr1.r1_penalty'].min()) > 1e-6      # the fixture itself
    assert rel_err(logits, g['Dr1.logits']) < tol_fwd
    # d logit / d img is a GRADIENT through 15 leaky-ReLU layers on unit-variance weights: a pre-activation within rounding
    # of zero takes the other slope under another fp32 evaluation order and changes the gradient at the pixels it feeds, so
    # single samples are held to TWICE the gradient tolerance (measured on MI355X, tools/diag_r1.py: split-bf16 2.8e-4, exact-fp32
    # MFMA 5.3e-4, fp16 x 3 1.05e-3 from the reference -- three fp32-class arithmetics, three sets of flipped slopes; any two differ
    # by 4e-3 at isolated pixels of the full tensor) while its L1 / L2 moments and the penalty -- where isolated pixels average
    # out -- meet the forward tolerance (measured 4e-6, 1.2e-5)
    _summary_ok(g, 'Dr1.r1_grads', gx, sample_slack * tol_grad, R1.SAMPLES)
    m, mg = PF.summarize(gx, samples=R1.SAMPLES)['moments'], g['Dr1.r1_grads.moments']
    assert abs(m[1] - mg[1]) <= tol_fwd * abs(mg[1]) and abs(m[2] - mg[2]) <= tol_fwd * abs(mg[2])
    assert rel_err(pen, g['Dr1.r1_penalty']) < tol_fwd
    checked = 0
    for k in FW.GRAD_KEYS_D:
        if 'Dr1.grad.' + k + '.sample' in g:
            _summary_ok(g, 'Dr1.grad.' + k, grads[k], tol_grad, R1.SAMPLES)
            checked += 1
    assert checked >= 14
    _gradnorms_ok(g, 'Dr1.gradnorms', grads, tol_grad, floor=1e-6)


@pytest.mark.timeout(600)
def test_oracle_discriminator_r1_phase_at_full_width():
    from oracle import ref_networks as RN
    from oracle import make_golden_fullwidth_r1 as R1
    from training import networks
    g = load_golden('models_fullwidth_r1.npz')
    m = PF.fill_module(networks.Discriminator(**FW.D_KWARGS), kind='normal')
    params = dict(m.named_parameters())
    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in list(m.named_parameters()) + list(m.named_buffers())}
    names = sorted(params)
    x, c = R1.d_inputs()
    logits, gx, pen, grads = R1.dreg_phase(lambda img: RN.discriminator(sd, img, c), x, [sd[k] for k in names])
    _r1_check(g, logits, gx, pen, dict(zip(names, grads)), TOL_FWD, TOL_GRAD)


# ---- the HIP path -------------------------------------------------------------------------------------------------------

def _generator_gradients(math):
    from training import networks
    from torch_utils.ops import conv2d_gradfix
    old, conv2d_gradfix.conv_math = conv2d_gradfix.conv_math, math
    try:
        G = PF.fill_module(networks.GeneratorFull(**FW.G_KWARGS)).cuda().train().requires_grad_(True)
        inp = {k: v.cuda() for k, v in PF.make_inputs(n=2, seed=0).items()}
        img, fin, par = G(*_g_args(inp), noise_mode='const')
        probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
        probe.backward()
        return G, inp, (img, fin, par), probe, {k: p.grad for k, p in G.named_parameters()}
    finally:
        conv2d_gradfix.conv_math = old


@pytest.mark.gpu
def test_hip_generator_at_full_width():
    g = load_golden('models_fullwidth.npz')
    G, inp, (img, fin, par), probe, grads = _generator_gradients('default')
    for key, t in [('G.img', img), ('G.finetune_img', fin), ('G.pred_parsing', par)]:
        _summary_ok(g, key, t, TOL_FWD)
    assert abs(probe.item() - float(g['G.probe'][0])) < TOL_FWD * abs(float(g['G.probe'][0]))
    # Gradients at 1e-3.  A key may exceed it only where fp32 arithmetic itself does: some weight gradients are what is left
    # after the demodulation cancels the bulk (b32.conv0.weight: entries of 1e-8 against 1e-3 elsewhere), and ANY fp32
    # evaluation order lands 1e-3 away from the CPU reference there.  The yardstick is this package's exact-fp32 mode
    # (fp32 MFMA, bit-for-bit fp32 FMA chains): the default split-bf16 arithmetic may deviate at most twice as far as it does.
    _, _, _, _, grads32 = _generator_gradients('f32')
    loosened = []
    for k in FW.GRAD_KEYS_G:
        ref = g['G.grad.' + k + '.sample']
        e32 = rel_err(PF.summarize(grads32[k], samples=FW.SAMPLES)['sample'], ref)
        tol = min(max(TOL_GRAD, 2 * e32), 1e-2)
        if tol > TOL_GRAD:
            loosened.append((k, e32))
        _summary_ok(g, 'G.grad.' + k, grads[k], tol)
    assert len(loosened) <= 2, loosened
    _gradnorms_ok(g, 'G.gradnorms', grads, TOL_GRAD)
    G.eval()
    with torch.no_grad():
        img_e, fin_e, _ = G(*_g_args(inp), noise_mode='const')        # eval: per-sample weights, grouped convolution
    _summary_ok(g, 'G.eval.img', img_e, TOL_FWD)
    _summary_ok(g, 'G.eval.finetune_img', fin_e, TOL_FWD)


@pytest.mark.gpu
def test_hip_discriminator_with_r1_at_full_width():
    from training import networks
    from torch_utils.ops import conv2d_gradfix
    g = load_golden('models_fullwidth.npz')
    D = PF.fill_module(networks.Discriminator(**FW.D_KWARGS)).cuda().train().requires_grad_(True)
    x, c = (t.cuda() for t in _d_inputs())
    x.requires_grad_(True)
    logits = D(x, c)
    assert rel_err(logits, g['D.logits']) < TOL_FWD
    with conv2d_gradfix.no_weight_gradients():
        gx, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    _summary_ok(g, 'D.r1_grads', gx, TOL_FWD)
    pen = gx.square().sum([1, 2, 3])
    assert rel_err(pen, g['D.r1_penalty']) < TOL_FWD
    (torch.nn.functional.softplus(-logits).mean() + 5.0 * pen.mean()).backward()
    grads = {k: p.grad for k, p in D.named_parameters()}
    for k in FW.GRAD_KEYS_D:
        _summary_ok(g, 'D.grad.' + k, grads[k], TOL_GRAD)
    _gradnorms_ok(g, 'D.gradnorms', grads, TOL_GRAD)


@pytest.mark.gpu
def test_hip_discriminator_r1_phase_at_full_width(arith):
    """The Dreg phase exactly as the loss runs it (loss_wo_flow_fullbody.py:236-254: r1_grads under no_weight_gradients,
    then backward through them): weight gradients of the input-gradient convolutions at 512 / 256 / 128 / 64 channels."""
    from oracle import make_golden_fullwidth_r1 as R1
    from training import networks
    from torch_utils.ops import conv2d_gradfix
    g = load_golden('models_fullwidth_r1.npz')
    D = PF.fill_module(networks.Discriminator(**FW.D_KWARGS), kind='normal').cuda().train().requires_grad_(True)
    names = sorted(dict(D.named_parameters()))
    sd = dict(D.named_parameters())
    x, c = (t.cuda() for t in R1.d_inputs())
    x = x.detach().requires_grad_(True)
    logits = D(x, c)
    with conv2d_gradfix.no_weight_gradients():
        gx, = torch.autograd.grad(outputs=[logits.sum()], inputs=[x], create_graph=True, only_inputs=True)
    pen = gx.square().sum([1, 2, 3])
    ((logits * 0 + pen * (R1.R1_GAMMA / 2)).mean() * R1.GAIN).backward()
    # split-bf16 (2.8e-4 measured) keeps the plain gradient tolerance on single samples of d logit / d img; the default gets the 2x of _r1_check
    _r1_check(g, logits, gx, pen, {k: sd[k].grad for k in names}, TOL_FWD, TOL_GRAD, sample_slack=2 if arith == 'f16x3' else 1)


@pytest.mark.gpu
def test_hip_discriminator_fp16_blocks():
    """num_fp16_res=4: b256..b32 in fp16 as the reference trains D (networks.py:1107, 1120; train_wo_flow_fullbody.py:195-196)."""
    from training import networks
    g = load_golden('models_fullwidth.npz')
    D = PF.fill_module(networks.Discriminator(**FW.D16_KWARGS)).cuda().train().requires_grad_(True)
    assert D.b256.use_fp16 and D.b32.use_fp16 and not D.b16.use_fp16
    x, c = (t.cuda() for t in _d_inputs())
    logits = D(x, c)
    assert logits.dtype == torch.float32 and rel_err(logits, g['D16.logits']) < TOL_FP16
    torch.nn.functional.softplus(-logits).mean().backward()
    grads = {k: p.grad for k, p in D.named_parameters()}
    for k in ['b256.conv0.weight', 'b64.conv1.weight', 'b16.conv1.weight', 'b4.fc.weight']:
        s = PF.summarize(grads[k], samples=FW.SAMPLES)
        assert rel_err(s['sample'], g['D16.grad.' + k + '.sample']) < 5 * TOL_FP16, k
    _gradnorms_ok(g, 'D16.gradnorms', grads, 5 * TOL_FP16, floor=1e-4)
