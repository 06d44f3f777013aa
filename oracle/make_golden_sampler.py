"""TEST INFRASTRUCTURE (build container only): index streams of the reference's ``misc.InfiniteSampler``
(torch_utils/misc.py:115-146) -> tests/golden/sampler.npz.  Imports the reference read-only; writes vectors, no source.

    python oracle/make_golden_sampler.py --ref /root/reference
"""

import argparse
import itertools
import json
import os
import sys

import numpy as np

CASES = [   # (dataset length, num_replicas, shuffle, seed, window_size, items drawn per rank)
    (37, 1, True, 0, 0.5, 200),
    (37, 4, True, 3, 0.5, 120),
    (8, 8, True, 1, 0.5, 40),
    (100, 2, True, 7, 0.02, 300),     # window of 2: the smallest that still swaps
    (100, 2, True, 7, 0.01, 150),     # window of 1: no swaps
    (11, 3, False, 0, 0.5, 50),
    (5, 2, True, 2, 1.0, 60),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'sampler.npz'))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    # harness-side accommodation (an ordinary TypeError, not a denial): the reference targets torch 1.7, whose Sampler.__init__
    # took the data source; torch 2.10's takes nothing
    import torch.utils.data
    torch.utils.data.Sampler.__init__ = lambda self, data_source=None: None
    from torch_utils import misc as ref_misc
    out, manifest = {}, []
    for k, (n, world, shuffle, seed, window, count) in enumerate(CASES):
        for rank in range(world):
            s = ref_misc.InfiniteSampler(list(range(n)), rank=rank, num_replicas=world, shuffle=shuffle, seed=seed, window_size=window)
            out[f'case{k}.rank{rank}'] = np.asarray(list(itertools.islice(iter(s), count)), dtype=np.int64)
        manifest.append(dict(n=n, world=world, shuffle=shuffle, seed=seed, window=window, count=count))
    out['manifest'] = np.asarray(json.dumps(manifest))
    np.savez_compressed(args.out, **out)
    print('wrote', args.out, len(out) - 1, 'streams')


if __name__ == '__main__':
    main()
