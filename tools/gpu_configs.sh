#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u tools/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err || { tail -20 gpurun_out/configs.err; exit 1; }
cat gpurun_out/configs.jsonl
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg -o cfg -- python3 tools/bench_configs.py --only 3,4,5 --reps 2 > gpurun_out/configs_rocprof.jsonl 2> gpurun_out/rocprof_cfg.err || { tail -20 gpurun_out/rocprof_cfg.err; exit 1; }
DB=$(ls gpurun_out/prof_cfg/*.db gpurun_out/prof_cfg/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/configs_kernel_stats.csv
