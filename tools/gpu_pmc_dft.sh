#!/bin/bash
# PMC passes (each its own run) for the frequency-domain stage kernels at the block-1 shape: MFMA busy + clock, FETCH_SIZE, WRITE_SIZE
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  local name=$1; local ctr=$2; shift 2
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d gpurun_out/pmc_$name -o p -- "$@" > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; exit 1; }
  DB=$(ls gpurun_out/pmc_$name/*.db gpurun_out/pmc_$name/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" dft_ > gpurun_out/pmc_$name.txt
  rm -rf gpurun_out/pmc_$name
}
run dft_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/dft_bench.py --reps 1 --no-direct
run dft_fetch "FETCH_SIZE" python3 tools/dft_bench.py --reps 1 --no-direct
run dft_write "WRITE_SIZE" python3 tools/dft_bench.py --reps 1 --no-direct
cat gpurun_out/pmc_dft_*.txt | cut -c1-170
