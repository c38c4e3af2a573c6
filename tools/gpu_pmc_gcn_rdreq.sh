#!/bin/bash
# read requests of the graph-conv kernels by size (TCC -> memory side): total, 32-byte, 64-byte, and DRAM reads in
# 32-byte units; the forward kernels (x read once with 16-byte-per-lane loads: 2.01 GB) calibrate the units
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 1 2; do
  rm -rf gpurun_out/gcr
  timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_32B_sum -d gpurun_out/gcr -o p -- python3 tools/gbench.py --block $b --only gcn --reps 1 > gpurun_out/gcr.log 2>&1 || { tail -5 gpurun_out/gcr.log; exit 1; }
  DB=$(ls gpurun_out/gcr/*.db gpurun_out/gcr/*/*.db 2>/dev/null | head -1)
  echo "== block $b"
  python tools/rocpd_summary.py pmc "$DB" gcn_ | sed 's/(anonymous namespace):://g' | cut -c1-150
  rm -rf gpurun_out/gcr
done
