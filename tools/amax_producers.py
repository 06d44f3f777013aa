"""Diagnostic: which operations produce the tensors that PASTA_MATH_F16X3 scans for their largest magnitude (bytes per step by producer)."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'pasta-gan_amd'), ROOT):
    sys.path.insert(0, p)
import torch
from torch_utils.ops import conv2d_gradfix as cg
from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch

stats = collections.defaultdict(lambda: [0, 0])
orig = cg.tensor_amax
def spy(t):
    hit = getattr(t, '_pasta_amax', None)
    if not (hit is not None and hit[0] == t._version and hit[1] == t.data_ptr()):
        if t.grad_fn is not None:
            who = type(t.grad_fn).__name__
        else:
            # no graph (backward pass or no_grad): name the caller chain instead
            fr = [f.name for f in traceback.extract_stack(limit=12)][:-2]
            who = 'nograd:' + '>'.join(n for n in fr if n not in ('apply', '_call_impl', '_wrapped_call_impl', 'forward', 'launch', '<module>'))[-90:]
        who += ' ' + str(tuple(t.shape))
        if isinstance(t, torch.nn.Parameter): who = 'Parameter'
        stats[who][0] += 1; stats[who][1] += t.numel() * 4
    return orig(t)
cg.tensor_amax = spy
dev = torch.device('cuda', 0)
step = TrainingStep(dev, batch_size=16, batch_gpu=16)
data = SyntheticFullBodyBatch(16, dev, seed=0)
step.run(data); step.run(data)
stats.clear()
step.run(data)           # a plain iteration (Gmain + Dmain)
tot = sum(v[1] for v in stats.values())
print('scanned %.2f GB in %d scans' % (tot / 1e9, sum(v[0] for v in stats.values())))
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%6.2f GB %4d scans  %s' % (v[1] / 1e9, v[0], k))
