"""Diagnostic: the merged discriminator pass run twice (bf16x6, f16x3); the inputs of corresponding convolution launches compared."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'pasta-gan_amd'), ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from oracle import param_fill as PF
from oracle.make_golden_loss import prepare
from training import networks
from training.loss_wo_flow_fullbody import StyleGAN2Loss
from torch_utils.ops import conv2d_gradfix as cg, bias_act as ba

G, D = prepare(networks.GeneratorFull(**PF.G_KWARGS).train(), networks.Discriminator(**PF.D_KWARGS).train())
D.cuda()
loss = StyleGAN2Loss(torch.device('cuda'), G.mapping, G.synthesis, G.const_encoding, G.style_encoding, D, vgg_weight=0, contextual_weight=0)
g = torch.Generator().manual_seed(9)
imgs = [(torch.rand([8, 3, 256, 256], generator=g) * 2 - 1).cuda().requires_grad_(True) for _ in range(3)]
cs = [torch.randn([8, 512], generator=g).cuda() for _ in range(3)]
w = [torch.randn([8, 1], generator=g).cuda() for _ in range(3)]
orig = cg._launch_conv
rec = {}
cur = None
def spy(x, wt, cfg, **kw):
    y = orig(x, wt, cfg, **kw)
    rec[cur].append((x.detach().clone(), y.detach().clone(), tuple(wt.shape), tuple(cfg[:4]), sorted(kw)))
    return y
cg._launch_conv = spy
for mode in ['bf16x6', 'default', 'f32']:
    cur = mode; rec[mode] = []
    cg.conv_math = mode
    mer = loss.run_D_multi(imgs, cs, sync=True)
    g_mer = torch.autograd.grad(sum((a * x).sum() for a, x in zip(mer, w)), imgs)
A, B, C = rec['bf16x6'], rec['default'], rec['f32']
for i in range(20, len(A)):
    xa, xb, xc = A[i][0].double(), B[i][0].double(), C[i][0].double()
    pm = xa.abs().amax(dim=[1, 2, 3])
    dab = (xa - xb).abs().amax(dim=[1, 2, 3]) / pm
    dac = (xa - xc).abs().amax(dim=[1, 2, 3]) / pm
    w = int(dab.argmax())
    print(i, tuple(A[i][0].shape), A[i][2], 'worst sample %d: f16x3-vs-bf16x6 %.2e, f32-vs-bf16x6 (same sample) %.2e (its worst %.2e @%d); sample max / tensor max %.2e' %
          (w, float(dab[w]), float(dac[w]), float(dac.max()), int(dac.argmax()), float(pm[w] / pm.max())))
