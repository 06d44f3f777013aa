"""Diagnostic: merged vs separate discriminator passes (tests/test_training_step_gpu.py) under each arithmetic, and every
forward-type launch of the merged pass run in both fp32-equivalent arithmetics on the same operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'pasta-gan_amd'), ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from oracle import param_fill as PF
from oracle.make_golden_loss import prepare
from training import networks
from training.loss_wo_flow_fullbody import StyleGAN2Loss
from torch_utils.ops import conv2d_gradfix as cg

def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
G, D = prepare(networks.GeneratorFull(**PF.G_KWARGS).train(), networks.Discriminator(**PF.D_KWARGS).train())
D.cuda()
loss = StyleGAN2Loss(torch.device('cuda'), G.mapping, G.synthesis, G.const_encoding, G.style_encoding, D, vgg_weight=0, contextual_weight=0)
g = torch.Generator().manual_seed(9)
imgs = [(torch.rand([8, 3, 256, 256], generator=g) * 2 - 1).cuda().requires_grad_(True) for _ in range(3)]
cs = [torch.randn([8, 512], generator=g).cuda() for _ in range(3)]
w = [torch.randn([8, 1], generator=g).cuda() for _ in range(3)]
res = {}
for mode in ['bf16x6', 'default', 'f32']:
    cg.conv_math = mode
    sep = [loss.run_D(i, c, sync=True) for i, c in zip(imgs, cs)]
    mer = loss.run_D_multi(imgs, cs, sync=True)
    g_sep = torch.autograd.grad(sum((a * x).sum() for a, x in zip(sep, w)), imgs)
    g_mer = torch.autograd.grad(sum((a * x).sum() for a, x in zip(mer, w)), imgs)
    res[mode] = (g_sep, g_mer)
    print(mode, 'merged vs separate:', [rel(b, a) for a, b in zip(g_sep, g_mer)], 'grad max', float(g_sep[0].abs().max()))
for mode in ['default', 'f32']:
    print(mode, 'vs bf16x6: separate', [rel(a, b) for a, b in zip(res[mode][0], res['bf16x6'][0])], 'merged', [rel(a, b) for a, b in zip(res[mode][1], res['bf16x6'][1])])

orig = cg._launch_conv
rows = []
def both(x, wt, cfg, **kw):
    cg.conv_math = 'bf16x6'; a = orig(x, wt, cfg, **kw)
    cg.conv_math = 'default'; b = orig(x, wt, cfg, **kw)
    d = (a.double() - b.double()).abs()
    # error relative to each output PLANE's own largest value (n, c): the local yardstick
    pm = a.double().abs().amax(dim=[2, 3], keepdim=True).clamp_min(1e-300)
    xa = x.abs().float(); am = float(xa.max())
    xm = xa.amax(dim=[1, 2, 3])
    rows.append((float((d / pm).max()), float(d.max() / a.abs().max()), tuple(x.shape), tuple(wt.shape), tuple(cfg[:4]), sorted(kw),
                 am, float(xm.min()), float((xa < am * 2.0 ** -28).float().mean()), float((xa == 0).float().mean())))
    return b
cg._launch_conv = both
mer = loss.run_D_multi(imgs, cs, sync=True)
g_mer = torch.autograd.grad(sum((a * x).sum() for a, x in zip(mer, w)), imgs)
rows.sort(key=lambda r: -r[0])
for r in rows[:16]:
    print('plane-rel %.2e tensor-rel %.2e x%s w%s cfg%s %s |x| max %.2e smallest per-sample max %.2e  frac<2^-28 amax %.3f zeros %.3f' % r)
