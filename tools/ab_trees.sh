#!/bin/bash
# Same-box A/B of two source trees (this one and an export of an earlier commit under _ab_old/, built in the container):
# per-kernel average durations under rocprofv3 for the convolution micro-benchmark.   bash tools/ab_trees.sh "<bench_conv args>"
set -e -o pipefail
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
ARGS=${1:---only fwd,dgrad --reps 20}
for T in old new old new; do
    D=$ROOT; [ $T = old ] && D=$ROOT/_ab_old
    rm -rf "$ROOT/gpurun_out/ab_$T"
    ( cd "$D" && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/ab_$T" -- python3 tools/bench_conv.py $ARGS > "$ROOT/gpurun_out/ab_$T.txt" 2>&1 )
    echo "== $T"
    python3 - "$ROOT/gpurun_out/ab_$T" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):6d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
    find "$ROOT/gpurun_out/ab_$T" -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
