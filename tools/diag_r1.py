"""Diagnostic: the full-width Dreg fixture (tests/golden/models_fullwidth_r1.npz) under both fp32 arithmetics of the HIP path.
Prints the deviation of logits, r1_grads, penalty and parameter-gradient norms from the reference's values."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'pasta-gan_amd'), ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import param_fill as PF, make_golden_fullwidth as FW, make_golden_fullwidth_r1 as R1
from training import networks
from torch_utils.ops import conv2d_gradfix

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'models_fullwidth_r1.npz'))
def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
res = {}
for math in ['default', 'f32']:
    conv2d_gradfix.conv_math = math
    D = PF.fill_module(networks.Discriminator(**FW.D_KWARGS), kind='normal').cuda().train().requires_grad_(True)
    names = sorted(dict(D.named_parameters())); sd = dict(D.named_parameters())
    x, c = (t.cuda() for t in R1.d_inputs())
    x = x.detach().requires_grad_(True)
    logits = D(x, c)
    with conv2d_gradfix.no_weight_gradients():
        gx, = torch.autograd.grad([logits.sum()], [x], create_graph=True)
    pen = gx.square().sum([1, 2, 3])
    ((logits * 0 + pen * 5).mean() * 16).backward()
    s = PF.summarize(gx, samples=R1.SAMPLES)
    gn = np.array([float(sd[k].grad.norm()) if sd[k].grad is not None else -1 for k in names])
    ref = g['Dr1.gradnorms']
    ok = ref > 1e-6 * ref.max()
    print(math, 'logits', rel(logits.detach().cpu().numpy(), g['Dr1.logits']), 'gx sample', rel(s['sample'], g['Dr1.r1_grads.sample']),
          'gx moments', (s['moments'] / g['Dr1.r1_grads.moments'] - 1), 'pen', rel(pen.detach().cpu().numpy(), g['Dr1.r1_penalty']),
          'gradnorm max rel', float(np.abs(gn[ok] / ref[ok] - 1).max()))
    for k in FW.GRAD_KEYS_D:
        if 'Dr1.grad.' + k + '.sample' in g:
            print('   ', k, rel(PF.summarize(sd[k].grad, samples=R1.SAMPLES)['sample'], g['Dr1.grad.' + k + '.sample']))
    res[math] = (gx.detach().clone(), {k: sd[k].grad.clone() for k in names if sd[k].grad is not None})
a, b = res['default'][0], res['f32'][0]
print('default vs f32: gx', float((a - b).abs().max() / b.abs().max()))
