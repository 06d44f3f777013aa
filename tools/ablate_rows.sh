#!/bin/bash
# Timing-only ablation of the row-reuse kernel (weights-in-registers schedule, PASTA_ROWS_PIPE=3): rebuilds the library with
# -DPASTA_ABLATE=<bits> and times the 256->128 3x3 layer at 128^2.  Results of ablated builds are garbage by design.
set -e
cd "$(dirname "$0")/.."
restore() {   # whatever happens (interrupt, failed compile, lease timeout): leave the default library behind
    python3 - <<PY
import sys; sys.path.insert(0, 'pasta-gan_amd')
from torch_utils import custom_ops
custom_ops.build(force=True)
PY
}
trap restore EXIT
for A in 0 1 2 3 4 7 8 15; do
    python3 - <<PY
import sys; sys.path.insert(0, 'pasta-gan_amd')
from torch_utils import custom_ops
custom_ops.build(force=True, extra_flags=['-DPASTA_ABLATE=$A'])
PY
    echo "PASTA_ABLATE=$A"
    PASTA_ROWS_PIPE=3 python3 tools/bench_conv.py --only fwd --match "spade 256" --reps 30 2>/dev/null | tail -1
done
