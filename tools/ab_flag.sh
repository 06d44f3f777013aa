#!/bin/bash
# Same-box A/B of a compile-time variant: builds the library with and without -D<flag> (the build digest covers the flags, so the
# default library is rebuilt at the end whatever happens) and times the given bench_conv.py selection under both.
#   bash tools/ab_flag.sh PASTA_WGRAD_WHOLE "--only wgrad --match spade"
set -e
cd "$(dirname "$0")/.."
FLAG=$1; shift
restore() { python3 -c "import sys; sys.path.insert(0, 'pasta-gan_amd'); from torch_utils import custom_ops; custom_ops.build()"; }
trap restore EXIT
for V in on off on off; do
    if [ $V = on ]; then python3 -c "import sys; sys.path.insert(0, 'pasta-gan_amd'); from torch_utils import custom_ops; custom_ops.build(extra_flags=['-D$FLAG'])";
    else python3 -c "import sys; sys.path.insert(0, 'pasta-gan_amd'); from torch_utils import custom_ops; custom_ops.build()"; fi
    echo "== $FLAG $V"
    python3 - "$V" "$FLAG" "$@" <<'PY'
import sys, os
sys.path.insert(0, 'pasta-gan_amd')
from torch_utils import custom_ops
v, flag = sys.argv[1], sys.argv[2]
# load the library that was just built with the matching flags (the stamp decides)
custom_ops.get_plugin(extra_flags=['-D' + flag] if v == 'on' else [])
sys.argv = ['bench_conv.py'] + sys.argv[3:]
sys.path.insert(0, 'tools')
import runpy
runpy.run_path('tools/bench_conv.py', run_name='__main__')
PY
done
