#!/bin/bash
# Socket power / clocks while one convolution shape runs back to back:  bash tools/power_probe.sh "spade 256"
# (read-only rocm-smi queries from a second process; the benchmark itself is tools/bench_conv.py)
M=${1:-spade 256}
OUT=gpurun_out/power_probe.txt
: > $OUT
python tools/bench_conv.py --only fwd --match "$M" --reps 30000 > gpurun_out/power_probe_bench.log 2>&1 &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction\|edge" >> $OUT
    echo "--" >> $OUT
    sleep 0.5
done
wait $BP
tail -2 gpurun_out/power_probe_bench.log >> $OUT
