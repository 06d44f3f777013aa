"""Achieved HBM bandwidth of the memory-bound kernels on their live shapes (batch 16, 'fashion' widths).
Algorithmic bytes = (elements read + elements written) * 4."""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))
import numpy as np
import torch
from torch_utils.ops import upfirdn2d, bias_act, fma
from training import networks


def timeit(fn, reps=20):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()      # every case starts from an empty caching allocator (round 4: a 67 MB output carved again and again out of the
    fn(); fn()                    # 539 MB blocks of the cases before it read 75 - 90 us for a 37 us kernel: [16,256,65,65])
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = torch.device('cuda')
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(dev)
    rows = []
    up_cases = [
        ('upfirdn blur pad1 g4', [16, 64, 257, 257], dict(padding=[1, 1, 1, 1], gain=4)),
        ('upfirdn blur pad2', [16, 64, 256, 256], dict(padding=[2, 2, 2, 2])),
        ('upfirdn blur pad1 g4', [16, 128, 129, 129], dict(padding=[1, 1, 1, 1], gain=4)),
        ('upfirdn blur pad1 g4', [16, 256, 65, 65], dict(padding=[1, 1, 1, 1], gain=4)),
        ('upfirdn blur pad1 g4', [16, 512, 33, 33], dict(padding=[1, 1, 1, 1], gain=4)),
        ('upfirdn blur pad1 g4', [16, 512, 9, 9], dict(padding=[1, 1, 1, 1], gain=4)),
        ('upfirdn down2', [16, 64, 256, 256], dict(down=2, padding=[1, 1, 1, 1])),
        ('upfirdn up2 (grad of down2)', [16, 64, 128, 128], dict(up=2, padding=[2, 1, 2, 1], gain=4)),
        ('upfirdn up2 rgb', [16, 3, 128, 128], dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    ]
    with torch.no_grad():
        for name, shape, kw in up_cases:
            x = torch.randn(shape, device=dev)
            y = upfirdn2d.upfirdn2d(x, f, **kw)
            ms = timeit(lambda: upfirdn2d.upfirdn2d(x, f, **kw))
            rows.append((name, shape, (x.numel() + y.numel()) * 4, ms))
        for shape in ([16, 64, 256, 256], [16, 128, 128, 128], [16, 512, 16, 16]):
            x = torch.randn(shape, device=dev); b = torch.randn([shape[1]], device=dev)
            ms = timeit(lambda: bias_act.bias_act(x, b, act='lrelu', gain=np.sqrt(2), clamp=256))
            rows.append(('bias_act lrelu fwd', shape, 2 * x.numel() * 4, ms))
            y = bias_act.bias_act(x, b, act='lrelu', gain=np.sqrt(2), clamp=256)
            ms = timeit(lambda: bias_act._launch(x, b, None, y, None, 1, 1, 3, 0.2, float(np.sqrt(2)), 256.0))
            rows.append(('bias_act lrelu grad', shape, 3 * x.numel() * 4, ms))
            ms = timeit(lambda: bias_act._bias_grad(x, 1))
            rows.append(('bias_grad (db)', shape, x.numel() * 4, ms))
        x = torch.randn([16, 128, 128, 128], device=dev); g = torch.randn_like(x); b = torch.randn_like(x)
        ms = timeit(lambda: networks.spade_modulate(x, g, b))
        rows.append(('spade_norm fwd', list(x.shape), 4 * x.numel() * 4, ms))
        s = torch.randn([16, 128], device=dev); nz = torch.randn([16, 1, 128, 128], device=dev)
        ms = timeit(lambda: fma.scale_planes(x, s, nz))
        rows.append(('scale_add (demod+noise)', list(x.shape), 2 * x.numel() * 4, ms))
        ms = timeit(lambda: fma.plane_dot(x, g))
        rows.append(('plane_dot', list(x.shape), 2 * x.numel() * 4, ms))
        # fused tails and the ADA kernels
        gbt = torch.randn([16, 256, 128, 128], device=dev)
        ms = timeit(lambda: networks.spade_modulate(x, gbt, None, relu_gain=1.2, clamp=256))
        rows.append(('spade_norm fwd (gamma|beta, relu)', list(x.shape), 4 * x.numel() * 4, ms))
        from torch_utils import misc
        grads = [torch.randn([n], device=dev) for n in [512 * 512 * 9] * 20 + [128 * 128 * 9] * 60 + [512] * 120]
        ms = timeit(lambda: misc.nan_to_num_(grads, nan=0, posinf=1e5, neginf=-1e5))
        rows.append(('nan_to_num over 200 tensors', [sum(t.numel() for t in grads)], 2 * sum(t.numel() for t in grads) * 4, ms))
        from torch_utils.ops import grid_sample_gradfix as gs
        from training.augment import _ColorAffine
        img = torch.randn([48, 3, 256, 256], device=dev); C = torch.randn([48, 4, 4], device=dev)
        ms = timeit(lambda: _ColorAffine.apply(img, C, 0))
        rows.append(('color_affine', list(img.shape), 2 * img.numel() * 4, ms))
        big = torch.randn([48, 3, 700, 700], device=dev)
        th = torch.tensor([[0.8, 0.3, 0.02], [-0.3, 0.8, -0.01]], device=dev).repeat(48, 1, 1)
        out = gs.affine_sample(big, th, (524, 524))
        ms = timeit(lambda: gs.affine_sample(big, th, (524, 524)))
        rows.append(('affine_sample fwd', list(out.shape), (out.numel() * 2) * 4, ms))           # reads ~ the sampled footprint
        dyo = torch.randn_like(out)
        ms = timeit(lambda: gs._AffineSampleAdjoint.apply(dyo, th, (700, 700)))
        rows.append(('affine_sample adjoint (gather)', list(big.shape), (big.numel() + out.numel()) * 4, ms))
        ms = timeit(lambda: x + g)
        rows.append(('torch add (reference point)', list(x.shape), 3 * x.numel() * 4, ms))
        ms = timeit(lambda: x.clone())
        rows.append(('torch copy (reference point)', list(x.shape), 2 * x.numel() * 4, ms))
        # the same at twice the size (537 MB: beyond what the 256 MB Infinity Cache keeps between repetitions)
        xb = torch.randn([16, 64, 256, 256], device=dev); gb_ = torch.randn_like(xb)
        ms = timeit(lambda: xb + gb_)
        rows.append(('torch add (reference point)', list(xb.shape), 3 * xb.numel() * 4, ms))
        ms = timeit(lambda: xb.clone())
        rows.append(('torch copy (reference point)', list(xb.shape), 2 * xb.numel() * 4, ms))
    print(f"{'kernel':34s} {'shape':24s} {'MB':>8s} {'us':>8s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
    for name, shape, nbytes, ms in rows:
        gbs = nbytes / ms / 1e6
        print(f'{name:34s} {str(shape):24s} {nbytes / 1e6:8.1f} {ms * 1e3:8.1f} {gbs:8.0f} {gbs / 8000:9.2f}')


if __name__ == '__main__':
    main()
