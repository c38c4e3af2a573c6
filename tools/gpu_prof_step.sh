#!/bin/bash
# rocprofv3 kernel summary of the fp32 train step (4 steps), written to gpurun_out/bench_kernel_stats.csv
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-kernels --no-bf16 --no-infer $BENCH_EXTRA > gpurun_out/bench_rocprof.json 2> gpurun_out/rocprof.err || { tail -20 gpurun_out/rocprof.err; exit 1; }
DB=$(ls gpurun_out/prof/*.db gpurun_out/prof/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/bench_kernel_stats.csv > gpurun_out/bench_kernel_stats.txt
rm -rf gpurun_out/prof
head -45 gpurun_out/bench_kernel_stats.txt
