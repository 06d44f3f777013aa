"""Probe: discriminator forward + backward at batch 16 twice against batch 32 once (fp32, fashion widths)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))
import torch, dnnlib
from training.training_loop_wo_flow_fullbody import fashion_config
cfg = fashion_config()
D = dnnlib.util.construct_class_by_name(**cfg.D_kwargs).cuda().train()
def run(n, reps, wgrad):
    D.requires_grad_(wgrad)
    img = torch.randn([n, 3, 256, 256], device='cuda', requires_grad=not wgrad)
    c = torch.randn([n, 512], device='cuda')
    for _ in range(2):
        D(img, c).sum().backward()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        D(img, c).sum().backward()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for wgrad in (False, True):
    a, b, c3 = run(16, 10, wgrad), run(32, 10, wgrad), run(48, 6, wgrad)
    print(f'weight grads {wgrad}: batch 16: {a:.2f} ms (x2 = {2*a:.2f}, x3 = {3*a:.2f}); batch 32: {b:.2f} ms; batch 48: {c3:.2f} ms')
