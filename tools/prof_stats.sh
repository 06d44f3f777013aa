#!/bin/bash
# One rocprofv3 kernel-trace pass over a full lazy-regularisation period (16 iterations) -> gpurun_out/<tag>_kernel_stats.csv (per step).
set -e -o pipefail
TAG=${1:-r4x}; shift || true
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
rm -rf "gpurun_out/${TAG}_prof_stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/${TAG}_prof_stats" -- python3 bench.py --steps 15 --warmup 1 --no-cpu-baseline --no-variants "$@" > "gpurun_out/${TAG}_prof_stats.log" 2>&1
find "gpurun_out/${TAG}_prof_stats" -name "*.csv" ! -name "*kernel_stats.csv" -delete
python3 - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/{tag}_prof_stats/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(f'gpurun_out/{tag}_kernel_stats.csv', 'w') as o:
    o.write('kernel,calls,total_ms,avg_us,min_us,max_us,percent,ms_per_step\n')
    for r in rows:
        t = float(r['TotalDurationNs'])
        o.write('"%s",%d,%.3f,%.1f,%.1f,%.1f,%.2f,%.2f\n' % (r['Name'], int(r['Calls']), t / 1e6, float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, 100 * t / tot, t / 1e6 / 16))
print('kernel time per step %.1f ms, launches per step %.0f' % (tot / 1e6 / 16, sum(int(r['Calls']) for r in rows) / 16))
PY
tail -1 "gpurun_out/${TAG}_prof_stats.log" | cut -c1-300
