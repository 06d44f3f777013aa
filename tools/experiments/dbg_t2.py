import sys
sys.path.insert(0, '/root/repo/pasta-gan_amd')
import torch
from torch_utils.ops import conv2d_gradfix as cg, _native
dev = torch.device('cuda')
torch.manual_seed(0)
for (n, ci, h, co, groups, isc) in [(2, 32, 32, 64, 1, False), (2, 48, 32, 96, 1, False), (1, 64, 64, 40, 1, False), (2, 32, 32, 128, 2, False), (2, 32, 32, 64, 1, True), (3, 128, 32, 256, 1, True), (9, 32, 64, 64, 1, False), (33, 16, 32, 64, 1, False), (5, 64, 16, 96, 1, False), (17, 32, 16, 64, 1, True), (2, 32, 48, 64, 1, False)]:
    x = torch.randn([n, ci, h, h], device=dev)
    w = torch.randn([ci, co // groups, 3, 3], device=dev) * 0.05
    s = (torch.rand([n, ci], device=dev) + 0.5) if isc else None
    cfg = cg._Cfg((True, 2, 0, 0, 0, 0, groups))
    y = cg._launch_conv(x, w, cfg, iscale=s)
    xr = x.double() * (s.double()[:, :, None, None] if isc else 1.0)
    ref = torch.nn.functional.conv_transpose2d(xr, w.double(), stride=2, groups=groups)
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    print((n, ci, h, co, groups, isc), tuple(y.shape), 'rel err', err)
