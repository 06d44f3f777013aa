"""Golden vectors of the reference's GeneratorFull / Discriminator at the FULL ``cfg=fashion`` widths
(``channel_base=16384``: 512 channels up to 32x32, 256 / 128 / 64 at 64 / 128 / 256) -- build container only, called by
``oracle/make_golden.py --only fullwidth``.  TEST INFRASTRUCTURE.

These are the shapes BASELINE config 2 executes (batch 2 instead of 16: the tile, row-reuse, split-K and weight-gradient
plans of the convolution kernels are chosen by channel counts and plane sizes, which are the benchmark's).  Stored per
tensor: 512 strided samples + three moments; per parameter: the norm of its gradient.  Also a fp16 discriminator run
(``num_fp16_res=4``: blocks b256..b32 compute in fp16, networks.py:1107, 1120), the storage type the reference trains D in.
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))
from oracle import param_fill as PF  # noqa: E402

G_KWARGS = dict(PF.G_KWARGS, synthesis_kwargs=dict(channel_base=16384, channel_max=512, conv_clamp=256))
D_KWARGS = dict(PF.D_KWARGS, channel_base=16384)
D16_KWARGS = dict(D_KWARGS, num_fp16_res=4)

# one or two parameters of every convolution plan the step uses at these widths (the style encoder's trunk below its last
# layer is left to the gradient-norm check: behind the instance norms its gradients are cancellation residue, 1e-6 of the rest)
GRAD_KEYS_G = ['synthesis.b4.conv1.weight', 'synthesis.b8.conv0.weight', 'synthesis.b16.conv1.weight', 'synthesis.b32.conv0.weight',
               'synthesis.b32.merge_conv.weight', 'synthesis.b64.conv0.weight', 'synthesis.b64.conv1.weight', 'synthesis.b64.conv0.affine.weight',
               'synthesis.b128.conv0.weight', 'synthesis.b128.conv1.weight', 'synthesis.b128.merge_conv.weight', 'synthesis.b256.conv0.weight',
               'synthesis.b256.conv1.weight', 'synthesis.b256.torgb.weight', 'synthesis.b256.torgb.m_weight1',
               'synthesis.spade_b128_1.conv.weight', 'synthesis.spade_b128_1.skip.weight', 'synthesis.spade_b128_2.spade0.conv_mlp.weight',
               'synthesis.spade_b128_2.spade0.conv_gamma.weight', 'synthesis.spade_b128_3.spade1.conv_beta.weight', 'synthesis.spade_b128_3.conv1.weight',
               'synthesis.spade_encoder.0.weight', 'synthesis.spade_encoder.2.conv1.weight', 'synthesis.spade_encoder.2.skip.weight',
               'synthesis.texture_b256.conv0.weight', 'synthesis.texture_b256.conv1.noise_strength', 'synthesis.texture_b256.merge_conv.weight',
               'const_encoding.model.0.weight', 'const_encoding.model.1.weight', 'const_encoding.model.3.weight', 'const_encoding.model.6.weight',
               'style_encoding.model.12.weight', 'style_encoding.fc.weight', 'style_encoding.feat_enc.0.weight',
               'style_encoding.feat_enc.2.weight', 'style_encoding.feat_enc.2.bias', 'mapping.fc0.weight']
GRAD_KEYS_D = ['b256.fromrgb.weight', 'b256.conv0.weight', 'b256.conv1.weight', 'b256.skip.weight', 'b128.conv0.weight', 'b128.conv1.weight',
               'b64.conv1.weight', 'b64.skip.weight', 'b32.conv0.weight', 'b16.conv1.weight', 'b8.conv0.weight', 'b8.skip.weight', 'b4.conv.weight',
               'b4.fc.weight', 'mapping.fc3.bias', 'b4.out.weight']
SAMPLES = 512


def put(out, key, t):
    s = PF.summarize(t, samples=SAMPLES)
    out[key + '.sample'] = s['sample']
    out[key + '.moments'] = s['moments']


def gradnorms(module):
    return np.array([p.grad.float().norm().item() if p.grad is not None else -1.0 for _, p in sorted(dict(module.named_parameters()).items())])


def run_discriminator(D, out, tag, keys):
    c = torch.tanh(PF.make_inputs(n=4, seed=1)['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
    x = PF.make_inputs(n=4, seed=1)['real_img'].requires_grad_(True)
    logits = D(x, c)
    out[tag + '.logits'] = logits.detach().float().numpy()
    gx, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    put(out, tag + '.r1_grads', gx)
    pen = gx.square().sum([1, 2, 3])
    out[tag + '.r1_penalty'] = pen.detach().numpy()
    loss = torch.nn.functional.softplus(-logits).mean() + 5.0 * pen.mean()
    loss.backward()
    sd = dict(D.named_parameters())
    for k in keys:
        put(out, tag + '.grad.' + k, sd[k].grad)
    out[tag + '.gradnorms'] = gradnorms(D)


def gen_fullwidth(ref_root, import_reference_networks):
    rn = import_reference_networks(ref_root)
    torch.manual_seed(0)
    inp = PF.make_inputs(n=2, seed=0)
    out = {}

    G = PF.fill_module(rn.GeneratorFull(**G_KWARGS)).train().requires_grad_(True)
    img, fin, par = G(inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                      inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    put(out, 'G.img', img); put(out, 'G.finetune_img', fin); put(out, 'G.pred_parsing', par)
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    probe.backward()
    out['G.probe'] = np.array([probe.item()])
    sd = dict(G.named_parameters())
    for k in GRAD_KEYS_G:
        put(out, 'G.grad.' + k, sd[k].grad)
    out['G.gradnorms'] = gradnorms(G)
    G.eval()
    with torch.no_grad():
        img_e, fin_e, _ = G(inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'],
                            inp['denorm_lower_input'], inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    put(out, 'G.eval.img', img_e); put(out, 'G.eval.finetune_img', fin_e)
    del G

    D = PF.fill_module(rn.Discriminator(**D_KWARGS)).train().requires_grad_(True)
    run_discriminator(D, out, 'D', GRAD_KEYS_D)
    del D

    # fp16 blocks: logits and first-order gradients only (R1 through fp16 CPU kernels is not a meaningful yardstick)
    D16 = PF.fill_module(rn.Discriminator(**D16_KWARGS)).train().requires_grad_(True)
    c = torch.tanh(PF.make_inputs(n=4, seed=1)['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
    x = PF.make_inputs(n=4, seed=1)['real_img']
    logits = D16(x, c)
    out['D16.logits'] = logits.detach().float().numpy()
    torch.nn.functional.softplus(-logits).mean().backward()
    sd = dict(D16.named_parameters())
    for k in ['b256.conv0.weight', 'b64.conv1.weight', 'b16.conv1.weight', 'b4.fc.weight']:
        put(out, 'D16.grad.' + k, sd[k].grad)
    out['D16.gradnorms'] = gradnorms(D16)

    np.savez_compressed(os.path.join(GOLDEN, 'models_fullwidth.npz'), **out)
    print('full-width fixtures written:', len(out), 'arrays')
