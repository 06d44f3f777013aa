"""Where the HOST time of a training step goes (python -m cProfile style, top functions by own time).

    python tools/host_profile.py [--steps 3] [--top 45]

The step is GPU-bound only as long as the host can issue a step's ~4000 launches faster than the GPU executes them
(bench.py reports the pure issue time as host_issue_ms_per_step); this lists what the issue time is made of."""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--top', type=int, default=45)
    ap.add_argument('--sort', default='tottime')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--act-dtype', default=None)
    args = ap.parse_args()
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
    dev = torch.device('cuda', 0)
    step = TrainingStep(dev, cfg=fashion_config(act_dtype=args.act_dtype, mbstd_group_size=min(args.batch, 4)), batch_size=args.batch, batch_gpu=args.batch)
    data = SyntheticFullBodyBatch(args.batch, dev, seed=0)
    for _ in range(2):
        step.run(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run(data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'unprofiled: issue {1000 * (t1 - t0) / args.steps:.1f} ms/step, complete {1000 * (t2 - t0) / args.steps:.1f} ms/step')
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(args.steps):
        step.run(data)
    prof.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(prof)
    st.sort_stats(args.sort)
    print(f'(profile of {args.steps} steps; divide by {args.steps})')
    st.print_stats(args.top)


if __name__ == '__main__':
    main()
