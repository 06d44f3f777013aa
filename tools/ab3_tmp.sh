#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
for T in mid new mid new; do
    D=$ROOT; [ $T = mid ] && D=$ROOT/_ab_mid
    rm -rf "$ROOT/gpurun_out/ab3_$T"
    ( cd "$D" && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/ab3_$T" -- python3 tools/bench_conv.py --only fwd,dgrad --reps 20 > "$ROOT/gpurun_out/ab3_$T.txt" 2>&1 )
    echo "== $T"
    python3 - "$ROOT/gpurun_out/ab3_$T" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    if 'rows2d' in r['Name'] or 'bf16x6_kernel<128, 128' in r['Name'] or 'rows_bf16x6' in r['Name']:
        print(f"{r['Name'][:100]:100s} calls {int(r['Calls']):6d} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
    find "$ROOT/gpurun_out/ab3_$T" -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
