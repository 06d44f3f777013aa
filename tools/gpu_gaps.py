"""Idle time of the GPU inside the training step, from a rocprofv3 kernel trace: python tools/gpu_gaps.py <kernel_trace.csv> [steps]
Sums the gaps between consecutive kernels (end of one to start of the next, same queue order) over the last `steps` iterations' worth of launches and
lists the kernels in front of which the GPU waited longest -- where the host, not the GPU, sets the pace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
tail = rows[n // 2:]                                    # second half of the run: steady state
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tail)
wall = int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])
gaps = collections.Counter(); cnt = collections.Counter(); big = 0
idle = 0
for a, b in zip(tail, tail[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if g > 0:
        idle += g
        k = b['Kernel_Name'][:70]
        gaps[k] += g; cnt[k] += 1
        if g > 20000: big += g
print(f'kernels {len(tail)}  wall {wall / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms  idle {idle / 1e6:.1f} ms ({100 * idle / wall:.1f} %)  in gaps > 20 us: {big / 1e6:.1f} ms')
for k, v in gaps.most_common(25):
    print(f'{v / 1e6:8.2f} ms  {cnt[k]:6d} gaps  avg {v / cnt[k] / 1e3:6.1f} us  before {k}')
