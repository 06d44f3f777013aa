#!/bin/bash
# Counters of one kernel of a micro-benchmark run: bash tools/pmc_kernel.sh <kernel substring> <bench_conv args...>
export TMPDIR=/tmp
K="$1"; shift
OUT=gpurun_out/pmck; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 tools/bench_conv.py "$@" > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/busy -- python3 tools/bench_conv.py "$@" > $OUT/busy.log 2>&1
python3 - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
for d in ('sq', 'busy'):
    f = glob.glob(f'gpurun_out/pmck/{d}/*/*counter_collection.csv')[0]
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for c in acc: print(f'{c:28s} {acc[c] / n[c]:16.0f}  per launch ({n[c]} launches)')
PY
