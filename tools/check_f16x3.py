"""Accuracy of PASTA_MATH_F16X3 against fp64 on every kernel family, next to bf16x6 and fp32 MFMA (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))
import torch
import torch.nn.functional as F
from torch_utils.ops import conv2d_gradfix as cg

def errs(a, ref):
    d = (a.double().cpu() - ref).abs()
    return float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())

CASES = [
    ('3x3 s1 128^2-like (rows2d wide)', [4, 256, 64, 64], [128, 256, 3, 3], dict(padding=1), 1.0),
    ('3x3 s1 scale 1e-3', [4, 256, 64, 64], [128, 256, 3, 3], dict(padding=1), 1e-3),
    ('3x3 s1 scale 300', [4, 256, 64, 64], [128, 256, 3, 3], dict(padding=1), 300.0),
    ('3x3 s1 scale 1e30', [4, 64, 32, 32], [64, 64, 3, 3], dict(padding=1), 1e30),
    ('3x3 s1 scale 1e-30', [4, 64, 32, 32], [64, 64, 3, 3], dict(padding=1), 1e-30),
    ('3x3 s1 64ch 256 tile', [2, 64, 128, 128], [64, 64, 3, 3], dict(padding=1), 1.0),
    ('3x3 s2 (base kernel)', [8, 64, 65, 65], [128, 64, 3, 3], dict(stride=2), 1.0),
    ('1x1', [8, 192, 32, 32], [128, 192, 1, 1], dict(), 1.0),
    ('3x3 width 24 (base)', [8, 64, 24, 24], [160, 64, 3, 3], dict(padding=1), 1.0),
    ('3x3 small plane split-K', [16, 512, 8, 8], [512, 512, 3, 3], dict(padding=1), 1.0),
    ('heavy tail x', [4, 128, 64, 64], [128, 128, 3, 3], dict(padding=1), -1.0),
]
for name, xs, ws, kw, scale in CASES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(xs, generator=g)
    if scale < 0:
        x[:, ::7, ::5, ::3] *= 1e4
    else:
        x = x * scale
    w = torch.randn(ws, generator=g) / (ws[1] * ws[2] * ws[3]) ** 0.5
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, **kw)
    dy = torch.randn(y64.shape, generator=g)
    rx, rw = torch.autograd.grad(y64, [x64, w64], dy.double())
    line = [name]
    for mode in ['f32', 'bf16x6', 'f16x3']:
        cg.conv_math = mode
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        y = cg.conv2d(xg, wg, **kw)
        gx, gw = torch.autograd.grad(y, [xg, wg], dy.cuda())
        e = [errs(y.detach(), y64.detach()), errs(gx, rx), errs(gw, rw)]
        line.append(mode + ' y %.1e/%.1e dx %.1e/%.1e dw %.1e/%.1e' % (e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1]))
    print(' | '.join(line), flush=True)
cg.conv_math = 'default'
# non-finite element stays local
cg.conv_math = 'f16x3'
g = torch.Generator().manual_seed(5)
x = torch.randn([2, 64, 32, 32], generator=g).cuda(); w = (torch.randn([64, 64, 3, 3], generator=g) / 24).cuda()
clean = cg.conv2d(x, w, padding=1)
for bad in [float('inf'), float('nan')]:
    xp = x.clone(); xp[1, 7, 10, 20] = bad
    y = cg.conv2d(xp, w, padding=1)
    hit = torch.zeros_like(y, dtype=torch.bool); hit[1, :, 9:12, 19:22] = True
    print('non-finite', bad, 'local:', bool((~torch.isfinite(y[hit])).all()), 'rest identical:', bool(torch.equal(y[~hit], clean[~hit])))
# transposed
for kw in [dict(stride=2, padding=1), dict(stride=2, padding=0)]:
    g = torch.Generator().manual_seed(9)
    x = torch.randn([4, 128, 32, 32], generator=g); w = torch.randn([128, 64, 3, 3], generator=g) / 34
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv_transpose2d(x64, w64, **kw)
    dy = torch.randn(y64.shape, generator=g)
    rx, rw = torch.autograd.grad(y64, [x64, w64], dy.double())
    out = ['transposed %s' % kw]
    for mode in ['bf16x6', 'f16x3']:
        cg.conv_math = mode
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        y = cg.conv_transpose2d(xg, wg, **kw)
        gx, gw = torch.autograd.grad(y, [xg, wg], dy.cuda())
        out.append(mode + ' y %.1e dx %.1e dw %.1e' % (errs(y.detach(), y64.detach())[1], errs(gx, rx)[1], errs(gw, rw)[1]))
    print(' | '.join(out))
