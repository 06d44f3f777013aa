"""Golden vectors of the reference's Discriminator at the full ``cfg=fashion`` widths for the lazy-regularisation (Dreg)
phase ALONE, with a non-degenerate image derivative -- build container only, called by ``oracle/make_golden.py --only
fullwidth_r1``.  TEST INFRASTRUCTURE.

Why a second full-width discriminator fixture: in ``models_fullwidth.npz`` the closed-form wave fill makes
D almost independent of its image (|d logit / d img| ~ 1e-11, R1 penalty 4e-22), so the R1 term contributes nothing to the
checked parameter gradients and the double backward (weight gradient of the input-gradient convolutions, at 512 / 256 / 128
channels) met the reference only through its first-order part.  Here D carries unit-variance weights
(``param_fill.fill_module(kind='normal')``: the statistics of the reference's initialisation), the loss is the Dreg phase's
alone -- loss_wo_flow_fullbody.py:236-254 with do_Dmain = False: ``(real_logits * 0 + r1_penalty * r1_gamma / 2).mean() * gain``,
r1_gamma = 10, gain = D_reg_interval = 16 -- so EVERY stored parameter gradient is a second derivative.
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))
from oracle import param_fill as PF  # noqa: E402
from oracle.make_golden_fullwidth import D_KWARGS, GRAD_KEYS_D  # noqa: E402

R1_GAMMA, GAIN, BATCH = 10.0, 16.0, 4
SAMPLES = 509       # a prime count: the stride (numel // 509) is not a multiple of the row length, so the samples walk across the columns


def put(out, key, t):
    s = PF.summarize(t, samples=SAMPLES)
    out[key + '.sample'] = s['sample']
    out[key + '.moments'] = s['moments']


def d_inputs():
    inp = PF.make_inputs(n=BATCH, seed=2)
    c = torch.tanh(inp['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512] * 8)
    return inp['real_img'], c


def dreg_phase(logits_of, x, params):
    """The Dreg phase on any discriminator callable; returns (logits, r1_grads, r1_penalty, parameter gradients)."""
    x = x.detach().requires_grad_(True)
    logits = logits_of(x)
    gx, = torch.autograd.grad(outputs=[logits.sum()], inputs=[x], create_graph=True, only_inputs=True)
    pen = gx.square().sum([1, 2, 3])
    loss = (logits * 0 + pen * (R1_GAMMA / 2)).mean() * GAIN
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    return logits, gx, pen, grads


def gen_fullwidth_r1(ref_root, import_reference_networks):
    rn = import_reference_networks(ref_root)
    torch.manual_seed(0)
    D = PF.fill_module(rn.Discriminator(**D_KWARGS), kind='normal').train().requires_grad_(True)
    x, c = d_inputs()
    names = sorted(dict(D.named_parameters()))
    sd = dict(D.named_parameters())
    logits, gx, pen, grads = dreg_phase(lambda img: D(img, c), x, [sd[k] for k in names])
    out = {'Dr1.logits': logits.detach().numpy(), 'Dr1.r1_penalty': pen.detach().numpy()}
    put(out, 'Dr1.r1_grads', gx)
    g = dict(zip(names, grads))
    for k in GRAD_KEYS_D:
        if g[k] is not None:
            put(out, 'Dr1.grad.' + k, g[k])
    out['Dr1.gradnorms'] = np.array([g[k].float().norm().item() if g[k] is not None else -1.0 for k in names])
    print('logits', logits.detach().flatten().tolist(), 'penalty', pen.detach().tolist(), 'max |d logit / d img|', float(gx.abs().max()))
    print('gradient norms: min %.3e median %.3e max %.3e' % tuple(np.quantile(out['Dr1.gradnorms'][out['Dr1.gradnorms'] >= 0], [0, 0.5, 1])))
    np.savez_compressed(os.path.join(GOLDEN, 'models_fullwidth_r1.npz'), **out)
    print('full-width R1 fixture written:', len(out), 'arrays')
