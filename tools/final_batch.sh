set -o pipefail
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r2 2>&1 | tail -8
python3 bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; tail -2 gpurun_out/r2_bench_final.err
python3 bench.py --mode infer --res 512 --steps 20 --warmup 3 > gpurun_out/r2_infer512_final.json 2>/dev/null
python3 bench.py --mode infer --res 256 --steps 20 --warmup 3 > gpurun_out/r2_infer256_final.json 2>/dev/null
python3 bench.py --storage bf16 --train-res 512 --batch-gpu 8 --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_bf16_512_final.json 2>/dev/null
python3 bench.py --storage bf16 --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_bf16_final.json 2>/dev/null
python3 tools/bench_hbm_ops.py > gpurun_out/r2_hbm_microbench.txt 2>/dev/null
python3 tools/bench_conv.py --reps 10 > gpurun_out/r2_conv_microbench.txt 2>/dev/null
./tools/microbench/mfma_shape > gpurun_out/r2_mfma_shape.txt 2>&1
python3 tools/phase_times.py > gpurun_out/r2_phase_times.txt 2>&1 || true
echo BATCH DONE
