# One gpurun call: five rocprofv3 passes, the bench lines, the micro-benchmarks, the power probe.   bash tools/final_batch.sh r5
set -o pipefail
TAG=${1:-r5}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
bash tools/profile_round.sh ${TAG} 2>&1 | tail -8
python3 bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err; tail -2 gpurun_out/${TAG}_bench_final.err
python3 bench.py --mode infer --res 512 --steps 20 --warmup 3 > gpurun_out/${TAG}_infer512_final.json 2>/dev/null
python3 bench.py --mode infer --res 256 --steps 20 --warmup 3 > gpurun_out/${TAG}_infer256_final.json 2>/dev/null
python3 bench.py --storage bf16 --train-res 512 --batch-gpu 8 --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16_512_final.json 2>/dev/null
python3 bench.py --storage bf16 --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16_final.json 2>/dev/null
python3 tools/bench_hbm_ops.py > gpurun_out/${TAG}_hbm_microbench.txt 2>/dev/null
python3 tools/bench_conv.py --reps 10 > gpurun_out/${TAG}_conv_microbench.txt 2>/dev/null      # (ends with the down-path section: fp32 tensor against operand pieces)
./tools/microbench/mfma_shape > gpurun_out/${TAG}_mfma_shape.txt 2>&1
python3 tools/phase_times.py > gpurun_out/${TAG}_phase_times.txt 2>&1 || true
echo BATCH1 DONE
# round-2 additions: the inference workload's per-kernel times, the chip's power / clock under the dominant kernel
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_infer512_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_infer512_stats -- python3 bench.py --mode infer --res 512 --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/${TAG}_infer512_stats.log 2>&1
find gpurun_out/${TAG}_infer512_stats -name "*.csv" ! -name "*kernel_stats.csv" -delete
bash tools/power_probe.sh "spade 256" > /dev/null 2>&1; cp gpurun_out/power_probe.txt gpurun_out/${TAG}_power_probe.txt
echo BATCH2 DONE
