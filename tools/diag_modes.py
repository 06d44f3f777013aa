"""Diagnostic: the reference fixtures' generator gradients (models_fullbody.npz) under the three fp32-class arithmetics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'pasta-gan_amd'), ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import param_fill as PF
from oracle.make_golden_models import GRAD_KEYS_G
from training import networks
from torch_utils.ops import conv2d_gradfix as cg

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'models_fullbody.npz'))
out = {}
for mode in ['f32', 'bf16x6', 'f16x3']:
    cg.conv_math = mode
    G = PF.fill_module(networks.GeneratorFull(**PF.G_KWARGS)).cuda().train().requires_grad_(True)
    inp = {k: v.cuda() for k, v in PF.make_inputs(n=2, seed=0).items()}
    args = (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'], inp['denorm_upper_mask'], inp['denorm_lower_mask'])
    img, fin, par = G(*args, noise_mode='const')
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    probe.backward()
    sd = dict(G.named_parameters())
    out[mode] = {k: rel(PF.summarize(sd[k].grad)['sample'], g['G.grad.' + k + '.sample']) for k in GRAD_KEYS_G}
    out[mode]['img'] = rel(PF.summarize(img)['sample'], g['G.img.sample'])
for k in list(GRAD_KEYS_G) + ['img']:
    print('%-55s f32 %.2e  bf16x6 %.2e  f16x3 %.2e' % (k, out['f32'][k], out['bf16x6'][k], out['f16x3'][k]))
# full gradient tensors of the two keys where the arithmetics differ most, for an offline comparison with an fp64 evaluation
save = {}
for mode in ['f32', 'bf16x6', 'f16x3']:
    cg.conv_math = mode
    G = PF.fill_module(networks.GeneratorFull(**PF.G_KWARGS)).cuda().train().requires_grad_(True)
    img, fin, par = G(*args, noise_mode='const')
    ((img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()).backward()
    sd = dict(G.named_parameters())
    for k in ['synthesis.b64.conv0.affine.weight', 'synthesis.b4.conv1.weight', 'synthesis.b64.conv0.weight', 'synthesis.b128.merge_conv.weight']:
        save[mode + ':' + k] = sd[k].grad.detach().cpu().numpy()
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'r3_grads_modes.npz'), **save)
