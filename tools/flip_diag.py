"""Diagnostic: per-sample difference of the merged and the separate discriminator passes over six seeds (profiles/r4_merged_d_flips.txt)."""
import sys, os
sys.path.insert(0, 'pasta-gan_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from training import networks
from oracle import param_fill as PF
from training.loss_wo_flow_fullbody import StyleGAN2Loss
import importlib
tm = importlib.import_module('test_training_step_gpu')
def run(seed):
    G, D = tm.prepare(networks.GeneratorFull(**PF.G_KWARGS).train(), networks.Discriminator(**PF.D_KWARGS).train())
    D.cuda()
    loss = StyleGAN2Loss(torch.device('cuda'), G.mapping, G.synthesis, G.const_encoding, G.style_encoding, D, vgg_weight=0, contextual_weight=0)
    g = torch.Generator().manual_seed(seed)
    imgs = [(torch.rand([8, 3, 256, 256], generator=g) * 2 - 1).cuda().requires_grad_(True) for _ in range(3)]
    cs = [torch.randn([8, 512], generator=g).cuda() for _ in range(3)]
    sep = [loss.run_D(i, c, sync=True) for i, c in zip(imgs, cs)]
    mer = loss.run_D_multi(imgs, cs, sync=True)
    w = [torch.randn([8, 1], generator=g).cuda() for _ in range(3)]
    g_sep = torch.autograd.grad(sum((a * x).sum() for a, x in zip(sep, w)), imgs)
    g_mer = torch.autograd.grad(sum((a * x).sum() for a, x in zip(mer, w)), imgs)
    out = []
    for a, b in zip(g_sep, g_mer):
        pm = a.abs().amax(dim=[1, 2, 3]).clamp_min(1e-300)
        out += ((b - a).abs().amax(dim=[1, 2, 3]) / pm).cpu().tolist()
    return out
for seed in (9, 1, 2, 3, 4, 5):
    ps = run(seed)
    print(os.environ.get('PASTA_SKIP_ADD_FUSED', '1'), os.environ.get('PASTA_GRAD_JOIN', '1'), 'seed', seed, 'flips>=3e-4:', sum(p >= 3e-4 for p in ps), 'worst %.2e' % max(ps), 'median %.1e' % sorted(ps)[12])
