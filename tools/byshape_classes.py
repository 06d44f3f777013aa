"""Aggregate bench.py --by-shape output (stderr) into operator classes: ms per step, calls, mean TFLOP/s.  python tools/byshape_classes.py <file> [-v]"""
import collections, re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\('(\w+)', (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)\s+calls/step=\s*([\d.]+) ms/step=\s*([\d.]+) TF/s=\s*([\d.]+)", l)
    if m:
        rows.append((m.group(1),) + tuple(map(int, m.groups()[1:10])) + tuple(map(float, m.groups()[10:])))
def cls(r):
    kind, n, ci, h, co, oh, k, st, tr, g, calls, ms, tf = r
    if kind == 'wgrad':
        return 'wgrad k%d s%d%s' % (k, st, '' if min(h, oh) > 16 else ' (<=16 px rows)')
    if k == 1: return '1x1' + (' few-channel' if min(ci, co) < 16 else '')
    if k == 3 and st == 2 and not tr: return 's2 fwd'
    if k == 3 and st == 2 and tr: return 'T s2 (in >= 128: pair)' if h >= 128 else 'T s2 small'
    if k == 3 and st == 1: return ('3x3 s1 >= 32 px' if oh >= 32 else '3x3 s1 < 32 px') + (' few-channel' if min(ci, co) < 16 else '')
    return 'other k%d' % k
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for r in rows:
    a = agg[cls(r)]; a[0] += r[11]; a[1] += r[10]; a[2] += r[11] * r[12]
print('total %.2f ms/step over %d shapes' % (sum(v[0] for v in agg.values()), len(rows)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%-28s %7.2f ms/step %6.1f calls  %6.1f TFLOP/s' % (k, v[0], v[1], v[2] / max(v[0], 1e-9)))
if len(sys.argv) > 2:
    for r in sorted(rows, key=lambda r: -r[11]):
        if sys.argv[2] in cls(r): print(r)
