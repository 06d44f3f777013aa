#!/bin/bash
# Same-box A/B of two builds of this tree: lib/libpasta_hip.so (default flags) against lib/libpasta_hip_ab.so (built in the container with
# other compile flags and copied there).   bash tools/ab_lib.sh micro "<bench_conv args>"   |   bash tools/ab_lib.sh step
cd "$(dirname "$0")/.."
AB=$PWD/pasta-gan_amd/lib/libpasta_hip_ab.so
for V in ab default ab default; do
    if [ $V = ab ]; then export PASTA_LIB_AB=$AB; else unset PASTA_LIB_AB; fi
    echo "== $V"
    if [ "$1" = micro ]; then python3 tools/bench_conv.py $2 2>&1 | grep -v "amdgpu.ids\|^shape"
    else python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', d['roofline']['achieved'], 'TFLOP/s')"; fi
done
