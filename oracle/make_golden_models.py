"""Model-level golden vectors from the reference's own GeneratorFull / Discriminator (build container only).
Called by ``oracle/make_golden.py --only models``."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))
from oracle import param_fill as PF  # noqa: E402

GRAD_KEYS_G = ['synthesis.b4.conv1.weight', 'synthesis.b64.conv0.affine.weight', 'synthesis.b256.torgb.m_weight1',
               'synthesis.spade_b128_2.spade0.conv_gamma.weight', 'synthesis.spade_encoder.0.weight',
               'synthesis.texture_b256.conv1.noise_strength', 'const_encoding.model.3.weight', 'style_encoding.model.12.weight', 'style_encoding.fc.weight',
               'style_encoding.feat_enc.2.bias', 'mapping.fc0.weight', 'synthesis.b128.merge_conv.weight']
GRAD_KEYS_D = ['b256.fromrgb.weight', 'b64.conv1.weight', 'b8.skip.weight', 'b4.conv.weight', 'b4.fc.weight', 'mapping.fc3.bias', 'b4.out.weight']


def put(out, key, t):
    s = PF.summarize(t)
    out[key + '.sample'] = s['sample']
    out[key + '.moments'] = s['moments']


def gen_models(ref_root, import_reference_networks):
    rn = import_reference_networks(ref_root)
    torch.manual_seed(0)
    inp = PF.make_inputs(n=2, seed=0)

    # ---- generator: forward (const noise, train mode => non-fused modconv) and gradients of a scalar probe.
    G = PF.fill_module(rn.GeneratorFull(**PF.G_KWARGS)).train().requires_grad_(True)
    out = {}
    img, fin, par = G(inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                      inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    put(out, 'G.img', img); put(out, 'G.finetune_img', fin); put(out, 'G.pred_parsing', par)
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    probe.backward()
    out['G.probe'] = np.array([probe.item()])
    sd = dict(G.named_parameters())
    for k in GRAD_KEYS_G:
        put(out, 'G.grad.' + k, sd[k].grad)
    out['G.gradnorms'] = np.array([p.grad.norm().item() if p.grad is not None else -1.0 for _, p in sorted(sd.items())])

    # ---- generator in eval mode (fused modconv, grouped convolution), forward only.
    G.eval()
    with torch.no_grad():
        img_e, fin_e, par_e = G(inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'],
                                inp['denorm_lower_input'], inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    put(out, 'G.eval.img', img_e); put(out, 'G.eval.finetune_img', fin_e)

    # ---- GeneratorV18 (test.py's class): eval mode = fused modconv / grouped convolution, 60-channel patch input.
    G18 = PF.fill_module(rn.GeneratorV18(**PF.G_KWARGS)).eval().requires_grad_(False)
    c60 = PF.make_inputs(n=2, seed=5)['style_input'].repeat(1, 2, 1, 1)[:, :60]
    with torch.no_grad():
        o18 = G18(inp['gen_z'], c60, inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                  inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    for name, t in zip(['img', 'finetune_img', 'upper_mask', 'lower_mask'], o18):
        put(out, 'G18.' + name, t)

    # ---- discriminator: logits, parameter gradients and the R1 double-backward path.
    D = PF.fill_module(rn.Discriminator(**PF.D_KWARGS)).train().requires_grad_(True)
    c = torch.tanh(PF.make_inputs(n=4, seed=1)['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
    x = PF.make_inputs(n=4, seed=1)['real_img'].requires_grad_(True)
    logits = D(x, c)
    out['D.logits'] = logits.detach().numpy()
    gx, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    put(out, 'D.r1_grads', gx)
    pen = gx.square().sum([1, 2, 3])
    out['D.r1_penalty'] = pen.detach().numpy()
    loss = torch.nn.functional.softplus(-logits).mean() + 5.0 * pen.mean()
    loss.backward()
    sdd = dict(D.named_parameters())
    for k in GRAD_KEYS_D:
        put(out, 'D.grad.' + k, sdd[k].grad)
    out['D.gradnorms'] = np.array([p.grad.norm().item() if p.grad is not None else -1.0 for _, p in sorted(sdd.items())])
    np.savez_compressed(os.path.join(GOLDEN, 'models_fullbody.npz'), **out)
    print('model fixtures written')
