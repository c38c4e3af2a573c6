#!/bin/bash
# PMC passes for the frequency-domain weight-gradient kernel: HBM fetch and L2 hit / miss, square (block 2: C = 64, T = 120,
# stride 1) and rectangular (block 1: the two phases of the stride-2 conv) tiles
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  local name=$1; local ctr=$2; shift 2
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d gpurun_out/pmc_$name -o p -- "$@" > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; exit 1; }
  DB=$(ls gpurun_out/pmc_$name/*.db gpurun_out/pmc_$name/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" dft_wgrad > gpurun_out/pmc_$name.txt
  rm -rf gpurun_out/pmc_$name
}
run dwg_sq_fetch "FETCH_SIZE" python3 tools/dft_bench.py --reps 1 --no-direct --only "bwd: wgrad" --stride 1 --t 120
run dwg_sq_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" python3 tools/dft_bench.py --reps 1 --no-direct --only "bwd: wgrad" --stride 1 --t 120
run dwg_rc_fetch "FETCH_SIZE" python3 tools/dft_bench.py --reps 1 --no-direct --only "bwd: wgrad"
run dwg_rc_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" python3 tools/dft_bench.py --reps 1 --no-direct --only "bwd: wgrad"
cat gpurun_out/pmc_dwg_*.txt | cut -c1-170
