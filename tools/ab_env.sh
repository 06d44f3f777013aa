#!/bin/bash
# Same-box A/B of an environment switch on the training step:  bash tools/ab_env.sh PASTA_MERGE_SPLIT 0 1   (alternates the two values twice)
VAR=$1; A=$2; B=$3; shift 3
cd "$(dirname "$0")/.."
for V in $A $B $A $B; do
    env $VAR=$V python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-variants "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$V', d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', d['roofline']['achieved'], 'TFLOP/s; conv total', d.get('conv_total', {}).get('ms_per_step'))"
done
