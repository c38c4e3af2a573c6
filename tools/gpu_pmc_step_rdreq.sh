#!/bin/bash
# memory-side read requests (DRAM reads in 32-byte units) and written bytes of EVERY kernel of one train step
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for ctr in TCC_EA0_RDREQ_DRAM_32B_sum WRITE_SIZE; do
  rm -rf gpurun_out/srq
  timeout -s KILL 200 rocprofv3 --pmc $ctr -d gpurun_out/srq -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-kernels --no-bf16 --no-infer > gpurun_out/srq.log 2>&1 || { tail -5 gpurun_out/srq.log; exit 1; }
  DB=$(ls gpurun_out/srq/*.db gpurun_out/srq/*/*.db 2>/dev/null | head -1)
  echo "== $ctr"
  python tools/rocpd_summary.py pmc "$DB" | sed 's/(anonymous namespace):://g' | cut -c1-170
  rm -rf gpurun_out/srq
done
