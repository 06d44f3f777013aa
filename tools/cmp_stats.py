"""Compare two per-kernel summaries (profiles/<tag>_kernel_stats.csv): ms per step by kernel, sorted by the difference."""
import csv, sys
def load(p):
    return {r['kernel']: (float(r['ms_per_step']), int(r['calls']), float(r['avg_us'])) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in set(a) | set(b):
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    rows.append((y[0] - x[0], k, x, y))
rows.sort(key=lambda r: -abs(r[0]))
print('total %.2f -> %.2f ms/step; launches %.0f -> %.0f per step' % (sum(v[0] for v in a.values()), sum(v[0] for v in b.values()), sum(v[1] for v in a.values()) / 16, sum(v[1] for v in b.values()) / 16))
for d, k, x, y in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print('%+7.2f ms  %7.2f -> %7.2f  calls %5d -> %5d  avg %7.1f -> %7.1f us  %s' % (d, x[0], y[0], x[1], y[1], x[2], y[2], k[:120]))
