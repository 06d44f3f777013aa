"""Which torch (ATen) element-wise kernels still run in the training step, on what shapes, and from where: one iteration of the
batch-16 step under torch.profiler with shapes and stacks, aggregated by (operator, input shapes, innermost frame of ours).
    python tools/aten_census.py [--iters 2] > gpurun_out/aten_census.txt
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))
sys.path.insert(0, ROOT)

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=1)
    ap.add_argument('--batch-gpu', type=int, default=16)
    args = ap.parse_args()
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
    dev = torch.device('cuda', 0)
    cfg = fashion_config(mbstd_group_size=min(args.batch_gpu, 4))
    step = TrainingStep(dev, cfg=cfg, num_gpus=1, rank=0, batch_size=args.batch_gpu, batch_gpu=args.batch_gpu)
    data = SyntheticFullBodyBatch(args.batch_gpu, dev, seed=0, res=256)
    for _ in range(17):            # past a lazy-regularisation iteration, so the profiled ones are plain
        step.run(data)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(args.iters):
            step.run(data)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, 'self_device_time_total', None)
        if dt is None:
            dt = getattr(ev, 'self_cuda_time_total', 0)
        if not dt or not ev.name.startswith('aten::'):
            continue
        where = ''
        for fr in (ev.stack or []):
            if 'pasta-gan_amd' in fr and 'custom_ops' not in fr:
                where = fr.split('pasta-gan_amd/')[-1]
                break
        if not where:
            where = 'autograd engine' if not ev.stack else (ev.stack[0][-60:])
        key = (ev.name, str([tuple(s) for s in (ev.input_shapes or []) if s]), where)
        agg[key][0] += 1
        agg[key][1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for v in agg.values())
    print(f'ATen kernels: {total / 1e3 / args.iters:.2f} ms per iteration, {sum(v[0] for v in agg.values()) / args.iters:.0f} launches')
    for (name, shapes, where), (n, us) in rows[:70]:
        print(f'{us / 1e3 / args.iters:7.3f} ms x{n / args.iters:5.1f}  {name:22s} {shapes[:90]:90s} {where[:90]}')


if __name__ == '__main__':
    main()
