set -o pipefail
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r4 2>&1 | tail -8
python3 bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; tail -2 gpurun_out/r4_bench_final.err
python3 bench.py --mode infer --res 512 --steps 20 --warmup 3 > gpurun_out/r4_infer512_final.json 2>/dev/null
python3 bench.py --mode infer --res 256 --steps 20 --warmup 3 > gpurun_out/r4_infer256_final.json 2>/dev/null
python3 bench.py --storage bf16 --train-res 512 --batch-gpu 8 --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/r4_bench_bf16_512_final.json 2>/dev/null
python3 bench.py --storage bf16 --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/r4_bench_bf16_final.json 2>/dev/null
python3 tools/bench_hbm_ops.py > gpurun_out/r4_hbm_microbench.txt 2>/dev/null
python3 tools/bench_conv.py --reps 10 > gpurun_out/r4_conv_microbench.txt 2>/dev/null
./tools/microbench/mfma_shape > gpurun_out/r4_mfma_shape.txt 2>&1
python3 tools/phase_times.py > gpurun_out/r4_phase_times.txt 2>&1 || true
echo BATCH1 DONE
# round-2 additions: the inference workload's per-kernel times, the chip's power / clock under the dominant kernel
export TMPDIR=/tmp
rm -rf gpurun_out/r4_infer512_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_infer512_stats -- python3 bench.py --mode infer --res 512 --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/r4_infer512_stats.log 2>&1
find gpurun_out/r4_infer512_stats -name "*.csv" ! -name "*kernel_stats.csv" -delete
bash tools/power_probe.sh "spade 256" > /dev/null 2>&1; cp gpurun_out/power_probe.txt gpurun_out/r4_power_probe.txt
echo BATCH2 DONE
