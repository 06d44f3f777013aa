"""Which framework (non-libpasta_hip) GPU kernels a training step launches, grouped by operator, input shapes and the
Python line that issued them -- the list of what is still worth fusing.

    python tools/aten_breakdown.py [--steps 2] [--top 40]

Uses torch.profiler (kineto over roctracer) on iterations 2.. of the training step; times are GPU kernel times."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--act-dtype', default=None)
    args = ap.parse_args()
    from torch.profiler import profile, ProfilerActivity
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
    dev = torch.device('cuda', 0)
    step = TrainingStep(dev, cfg=fashion_config(act_dtype=args.act_dtype, mbstd_group_size=min(args.batch, 4)), batch_size=args.batch, batch_gpu=args.batch)
    data = SyntheticFullBodyBatch(args.batch, dev, seed=0)
    for _ in range(2):
        step.run(data)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(args.steps):
            step.run(data)
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    for ev in prof.events():
        kernels = [k for k in (getattr(ev, 'kernels', None) or []) if 'pasta::' not in k.name]
        if not kernels:
            continue
        us = sum(k.duration for k in kernels)
        stack = ev.stack or []
        where = next((s for s in stack if 'pasta-gan_amd' in s and 'torch_utils/ops' not in s), stack[0] if stack else '?')
        where = where.replace(ROOT + '/', '')
        key = (ev.name, str(ev.input_shapes)[:90], where[-70:])
        rows[key][0] += len(kernels)
        rows[key][1] += us
        total += us
    print(f'framework kernels: {total / 1000 / args.steps:.2f} ms of GPU time per step, {sum(v[0] for v in rows.values()) / args.steps:.0f} launches per step')
    bysite = collections.defaultdict(lambda: [0, 0.0])
    for (name, shapes, where), (count, us) in rows.items():
        bysite[where][0] += count; bysite[where][1] += us
    print('--- by Python site (launches per step)')
    for where, (count, us) in sorted(bysite.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f'{count / args.steps:7.1f} launches {us / 1000 / args.steps:8.3f} ms/step  {where}')
    print('--- by operator and shape (time)')
    for key, (count, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f'{us / 1000 / args.steps:8.3f} ms/step {count / args.steps:7.1f} calls  {key[0]:<28} {key[1]:<92} {key[2]}')


if __name__ == '__main__':
    main()
