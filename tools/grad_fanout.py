"""Which tensors of the training step have more than one consumer in the autograd graph -- each costs one gradient addition
(three passes over the tensor) in the backward: walks the graph of every ``backward()`` of one iteration and lists (shape, producer node,
consumer nodes), largest first.
    python tools/grad_fanout.py > gpurun_out/grad_fanout.txt
"""
import collections
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))
sys.path.insert(0, ROOT)

import torch


def census(root, out):
    seen, stack = set(), [root.grad_fn]
    incoming = collections.defaultdict(list)
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        for nxt, idx in fn.next_functions:
            if nxt is None:
                continue
            incoming[(nxt, idx)].append(type(fn).__name__)
            stack.append(nxt)
    for (fn, idx), parents in incoming.items():
        if len(parents) < 2 or type(fn).__name__ == 'AccumulateGrad':
            continue
        try:
            shape = tuple(fn._input_metadata[idx].shape)
        except Exception:  # noqa: BLE001
            shape = ()
        out.append((math.prod(shape) if shape else 0, shape, type(fn).__name__, sorted(parents)))


def main():
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
    dev = torch.device('cuda', 0)
    bg = 16
    cfg = fashion_config(mbstd_group_size=4)
    step = TrainingStep(dev, cfg=cfg, num_gpus=1, rank=0, batch_size=bg, batch_gpu=bg)
    data = SyntheticFullBodyBatch(bg, dev, seed=0, res=256)
    step.run(data)
    rows = []
    real_backward = torch.Tensor.backward

    def spy(self, *a, **k):
        found = []
        census(self, found)
        rows.append(found)
        return real_backward(self, *a, **k)
    torch.Tensor.backward = spy
    try:
        step.run(data)
    finally:
        torch.Tensor.backward = real_backward
    for i, found in enumerate(rows):
        found.sort(key=lambda r: -r[0])
        mb = sum(r[0] * (len(r[3]) - 1) for r in found) * 4 * 3 / 1e6
        print(f'--- backward {i}: {len(found)} multi-consumer tensors, {sum(len(r[3]) - 1 for r in found)} additions, {mb:.0f} MB of addition traffic')
        for n, shape, prod, parents in found[:40]:
            print(f'  {str(shape):24s} {prod:34s} <- {", ".join(parents)}')


if __name__ == '__main__':
    main()
