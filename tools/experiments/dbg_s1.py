import sys, os
sys.path.insert(0, '/root/repo/pasta-gan_amd')
import torch
from torch_utils.ops import conv2d_gradfix as cg, _native
dev = torch.device('cuda')
torch.manual_seed(0)
n, ci, h, co = 2, 32, 32, 128
cfg = cg._Cfg((False, 1, 1, 1, 0, 0, 1))
w = torch.zeros([co, ci, 3, 3], device=dev)
for c in range(ci):
    w[c, c, 1, 1] = 1.0
for (nn, c, yy, xx) in [(0, 0, 5, 7), (0, 3, 5, 7), (0, 9, 0, 0), (1, 17, 31, 31), (0, 31, 16, 1)]:
    x = torch.zeros([n, ci, h, h], device=dev)
    x[nn, c, yy, xx] = 1.0
    _native.amax_attach(x, cg.tensor_amax(x))
    pieces, bound, shape = cg.pieces_pack(x)
    y0 = cg._launch_conv(x, w, cfg)
    y1 = cg._launch_conv(pieces, w, cfg, pieces=(bound, shape))
    print((nn, c, yy, xx), 'fp32 ->', y0.nonzero().tolist()[:6], y0.sum().item(), ' pieces ->', y1.nonzero().tolist()[:6], y1.sum().item())
