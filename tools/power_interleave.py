"""Is the training step energy-bound?  The dominant convolution alone in a loop, then interleaved with a memory-bound copy of growing size:
if the socket's power cap averages over more than a kernel, the convolution behind a low-power kernel runs faster than in the pure loop."""
import sys
sys.path.insert(0, '/root/repo/pasta-gan_amd')
import torch
from torch_utils.ops import conv2d_gradfix as cg, _native
dev = torch.device('cuda')
x = torch.randn([16, 256, 128, 128], device=dev); w = torch.randn([128, 256, 3, 3], device=dev) * 0.05
_native.amax_attach(x, cg.tensor_amax(x))
cfg = cg._Cfg((False, 1, 1, 1, 0, 0, 1))
src = torch.randn([64 * 1024 * 1024], device=dev); dst = torch.empty_like(src)
def run(copy_elems, reps=60):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    s0 = torch.cuda.Event(enable_timing=True); e0 = torch.cuda.Event(enable_timing=True)
    for _ in range(10): cg._launch_conv(x, w, cfg)
    torch.cuda.synchronize(); s0.record()
    for a, b in ev:
        if copy_elems: dst[:copy_elems].copy_(src[:copy_elems])
        a.record(); cg._launch_conv(x, w, cfg); b.record()
    e0.record(); torch.cuda.synchronize()
    conv = sum(a.elapsed_time(b) for a, b in ev[10:]) / (reps - 10)
    return conv, s0.elapsed_time(e0) / reps
for n in (0, 4 << 20, 16 << 20, 64 << 20):
    c, t = run(n)
    print(f'copy {n * 8 / 1e6:7.0f} MB between convolutions: conv {c * 1e3:7.1f} us, loop period {t * 1e3:7.1f} us')
