#!/bin/bash
# HBM traffic of the graph-conv kernels (forward with statistics, backward data / weight) at the block 1 / 2 shapes:
# FETCH_SIZE and WRITE_SIZE, one counter per run
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 1 2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/gct
    timeout -s KILL 120 rocprofv3 --pmc $ctr -d gpurun_out/gct -o p -- python3 tools/gbench.py --block $b --only gcn --reps 1 > gpurun_out/gct.log 2>&1 || { tail -5 gpurun_out/gct.log; exit 1; }
    DB=$(ls gpurun_out/gct/*.db gpurun_out/gct/*/*.db 2>/dev/null | head -1)
    echo "== block $b $ctr"
    python tools/rocpd_summary.py pmc "$DB" gcn_ | sed 's/(anonymous namespace):://g' | cut -c1-150
    rm -rf gpurun_out/gct
  done
done
