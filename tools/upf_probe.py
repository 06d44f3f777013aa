import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pasta-gan_amd'))
from torch_utils.ops import upfirdn2d
x = torch.randn([16, 64, 256, 256], device='cuda')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
f4 = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
f3 = upfirdn2d.setup_filter([1, 2, 1]).cuda()
f1 = upfirdn2d.setup_filter([1]).cuda() if False else None
print('4x4 pad2 (256->257)', timeit(lambda: upfirdn2d.upfirdn2d(x, f4, padding=[2, 2, 2, 2])))
print('4x4 pad (2,1) (256->256)', timeit(lambda: upfirdn2d.upfirdn2d(x, f4, padding=[2, 1, 2, 1])))
print('3x3 pad1 (256->256)', timeit(lambda: upfirdn2d.upfirdn2d(x, f3, padding=[1, 1, 1, 1])))
print('3x3 pad (2,1): 256->257', timeit(lambda: upfirdn2d.upfirdn2d(x, f3, padding=[2, 1, 2, 1])))
x7 = torch.randn([16, 64, 257, 257], device='cuda')
print('4x4 pad1 (257->257... out 256?)', timeit(lambda: upfirdn2d.upfirdn2d(x7, f4, padding=[1, 1, 1, 1])), upfirdn2d.upfirdn2d(x7, f4, padding=[1, 1, 1, 1]).shape)
print('4x4 pad (2,1) on 257 -> 257', timeit(lambda: upfirdn2d.upfirdn2d(x7, f4, padding=[2, 1, 2, 1])))
print('clone 256', timeit(lambda: x.clone()))
