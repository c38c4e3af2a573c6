#!/bin/bash
# per-kernel times of one bench_configs.py entry under rocprofv3:  CFG=4 bash tools/gpu_cfg_prof.sh
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
CFG=${CFG:-4}
rm -rf gpurun_out/prof_c$CFG
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c$CFG -o p -- python3 tools/bench_configs.py --only $CFG --reps 3 > gpurun_out/cfg${CFG}_prof.jsonl 2> gpurun_out/cfg${CFG}_prof.err || { tail -5 gpurun_out/cfg${CFG}_prof.err; exit 1; }
DB=$(ls gpurun_out/prof_c$CFG/*.db gpurun_out/prof_c$CFG/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/cfg${CFG}_kernel_stats.csv | head -${LINES_OUT:-30}
rm -rf gpurun_out/prof_c$CFG
