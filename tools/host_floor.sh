#!/bin/bash
# Host floor of the training step: at batch 2 the GPU work of a step is an eighth of batch 16's while the host issues the same launches,
# so the step time IS the host's.  Three conditions: all cores of the box; two cores and OMP_NUM_THREADS=2 (what one of
# eight ranks gets on a 16-core slice); one core.
cd "$(dirname "$0")/.."
run() {
    "$@" python3 bench.py --batch-gpu 2 --steps 16 --warmup 3 --no-cpu-baseline --no-variants --no-meter 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  ms/step', d['ms_per_step'], ' host issue ms/step', d.get('host_issue_ms_per_step'))"
}
echo "cores visible: $(nproc)"
echo "all cores:"; run env
echo "2 cores, OMP_NUM_THREADS=2:"; run env OMP_NUM_THREADS=2 taskset -c 0,1
echo "1 core, OMP_NUM_THREADS=1:"; run env OMP_NUM_THREADS=1 taskset -c 0
echo "batch 16, 2 cores, OMP_NUM_THREADS=2 (the condition of one rank of eight):"
env OMP_NUM_THREADS=2 taskset -c 0,1 python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  ms/step', d['ms_per_step'], ' img/s', d['value'])"
