#!/bin/bash
# rocprofv3 evidence for one round, on the GPU box:   bash tools/profile_round.sh r2
# Writes raw traces under gpurun_out/ (scratch); `python tools/summarize_profiles.py r2 --stats gpurun_out/r2_prof_stats
# --pmc gpurun_out --pmc-prefix r2pmc_ --steps 16` condenses them into profiles/.  Counters are collected in passes of
# their own (never together with a trace), the program itself follows `--` (no env / bash -c hop).
set -e -o pipefail
TAG=${1:-r3}
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
OUT=$ROOT/gpurun_out
rm -rf "$OUT/${TAG}_prof_stats" "$OUT/${TAG}pmc_FETCH_SIZE" "$OUT/${TAG}pmc_WRITE_SIZE" "$OUT/${TAG}pmc_SQ_VALU_MFMA_BUSY_CYCLES" "$OUT/${TAG}pmc_SQ"
# 1. per-kernel time over one full lazy-regularisation period (16 iterations = 1 warm-up + 15 timed)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof_stats" -- python3 bench.py --steps 15 --warmup 1 --no-cpu-baseline --no-variants > "$OUT/${TAG}_prof_stats.log" 2>&1
echo "stats pass done"
# 2-4. HBM traffic and matrix-pipe utilisation: iteration 0 (every phase) + one plain iteration
PMC_CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-meter --no-variants"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}pmc_FETCH_SIZE" -- $PMC_CMD > "$OUT/${TAG}pmc_fetch.log" 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}pmc_WRITE_SIZE" -- $PMC_CMD > "$OUT/${TAG}pmc_write.log" 2>&1
echo "WRITE_SIZE pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}pmc_SQ_VALU_MFMA_BUSY_CYCLES" -- $PMC_CMD > "$OUT/${TAG}pmc_busy.log" 2>&1
echo "MFMA busy pass done"
# 5. where the waves' cycles go
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d "$OUT/${TAG}pmc_SQ" -- $PMC_CMD > "$OUT/${TAG}pmc_sq.log" 2>&1
echo "SQ pass done"
# keep the merge-back small: the raw counter files are tens of MB each
for d in "$OUT/${TAG}pmc_FETCH_SIZE" "$OUT/${TAG}pmc_WRITE_SIZE" "$OUT/${TAG}pmc_SQ_VALU_MFMA_BUSY_CYCLES" "$OUT/${TAG}pmc_SQ" "$OUT/${TAG}_prof_stats"; do
    find "$d" -name "*.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete
    find "$d" -name "*agent_info*" -delete 2>/dev/null || true
done
python3 tools/summarize_profiles.py "$TAG" --stats "$OUT/${TAG}_prof_stats" --pmc "$OUT" --pmc-prefix "${TAG}pmc_" --steps 16 --sq "$OUT/${TAG}pmc_SQ" > "$OUT/${TAG}_summary.log" 2>&1 || cat "$OUT/${TAG}_summary.log"
mkdir -p "$OUT/${TAG}_profiles" && cp profiles/${TAG}_* "$OUT/${TAG}_profiles/" 2>/dev/null || true
tail -3 "$OUT/${TAG}_summary.log"
