"""GPU time of each training phase (Gmain, Greg, Dmain, Dreg, optimizer + EMA) over one lazy-regularisation period."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))
import torch
from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
from torch_utils import misc

dev = torch.device('cuda')
step = TrainingStep(dev, cfg=fashion_config(), batch_size=16, batch_gpu=16)
data = SyntheticFullBodyBatch(16, dev)
acc = {}
orig = step.loss.accumulate_gradients
def timed(phase, **kw):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); orig(phase=phase, **kw); e.record()
    acc.setdefault(phase, []).append((s, e))
step.loss.accumulate_gradients = timed
for _ in range(2):
    step.run(data)
torch.cuda.synchronize(); acc.clear()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(16):
    step.run(data)
t1.record(); torch.cuda.synchronize()
total = t0.elapsed_time(t1) / 16
print(f'step {total:.1f} ms')
s = 0
for k, v in acc.items():
    ms = sum(a.elapsed_time(b) for a, b in v) / 16
    s += ms
    print(f'{k:6s} calls/16it {len(v):3d}  {ms:7.2f} ms/step  ({sum(a.elapsed_time(b) for a, b in v) / len(v):7.2f} ms per call)')
print(f'optimizer steps + nan_to_num + EMA + randn: {total - s:.2f} ms/step')
