#!/bin/bash
# PMC passes (each its own run) for the five largest stage kernels of the frequency-domain temporal conv at the block-1
# shape (C = 64, T = 240, stride 2, N = 4096): MFMA busy + clock, FETCH_SIZE, WRITE_SIZE; then tools/pmc_traffic.py
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
for st in tr ib mf mb wg; do
  case $st in tr) only="fwd: transform";; ib) only="bwd: inverse";; mf) only="fwd: mix";; mb) only="bwd: mix";; wg) only="bwd: wgrad";; esac
  run dft_${st}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/dft_bench.py --reps 1 --no-direct --only "$only"
  run dft_${st}_fetch "FETCH_SIZE" python3 tools/dft_bench.py --reps 1 --no-direct --only "$only"
  run dft_${st}_write "WRITE_SIZE" python3 tools/dft_bench.py --reps 1 --no-direct --only "$only"
done
cat gpurun_out/pmc_dft_*.txt | cut -c1-170
