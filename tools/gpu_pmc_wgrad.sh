#!/bin/bash
# PMC passes for the dominant kernel of the step (stride-2 weight gradient, block 1 shape), each in its own run
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  local name=$1; local ctr=$2; shift 2
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d gpurun_out/pmc_$name -o p -- "$@" > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; exit 1; }
  DB=$(ls gpurun_out/pmc_$name/*.db gpurun_out/pmc_$name/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" tcn_ > gpurun_out/pmc_$name.txt
  rm -rf gpurun_out/pmc_$name
}
run wg_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/kbench.py all --n 4096 --c 64 --stride 2 --t 240 --reps 1
run wg_fetch "FETCH_SIZE" python3 tools/kbench.py all --n 4096 --c 64 --stride 2 --t 240 --reps 1
run wg_write "WRITE_SIZE" python3 tools/kbench.py all --n 4096 --c 64 --stride 2 --t 240 --reps 1
cat gpurun_out/pmc_wg_*.txt | cut -c1-170
