#!/bin/bash
# per-kernel times of the headline train step in the opt-in f32_split mode:  bash tools/gpu_split_prof.sh
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_split
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_split -o p -- python3 bench.py --mfma ${MODE:-f32_split_gcn} --steps 3 --warmup 2 --no-cpu --no-kernels > gpurun_out/split_prof.json 2> gpurun_out/split_prof.err || { tail -5 gpurun_out/split_prof.err; exit 1; }
DB=$(ls gpurun_out/prof_split/*.db gpurun_out/prof_split/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/split_kernel_stats.csv | head -${LINES_OUT:-40}
rm -rf gpurun_out/prof_split
