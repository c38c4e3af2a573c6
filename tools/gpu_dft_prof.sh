#!/bin/bash
# per-kernel times of the f32_dft train step under rocprofv3
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_dft
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dft -o p -- python3 bench.py --mfma ${MODE:-f32_dft} --steps 3 --warmup 1 --no-cpu --no-kernels --no-infer --no-bf16 > gpurun_out/bench_dft_prof.json 2> gpurun_out/bench_dft_prof.err || { tail -5 gpurun_out/bench_dft_prof.err; exit 1; }
DB=$(ls gpurun_out/prof_dft/*.db gpurun_out/prof_dft/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/bench_dft_kernel_stats.csv | head -48
rm -rf gpurun_out/prof_dft
