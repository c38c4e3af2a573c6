#!/bin/bash
# per-kernel times of the bf16-operand train step (bench.py --mfma bf16) under rocprofv3
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_bf16
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bf16 -o p -- python3 bench.py --mfma bf16 --steps 3 --warmup 1 --no-cpu --no-kernels --no-infer --no-bf16 > gpurun_out/bench_bf16_prof.json 2> gpurun_out/bench_bf16_prof.err || { tail -5 gpurun_out/bench_bf16_prof.err; exit 1; }
DB=$(ls gpurun_out/prof_bf16/*.db gpurun_out/prof_bf16/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/bench_bf16_kernel_stats.csv | head -45
rm -rf gpurun_out/prof_bf16
