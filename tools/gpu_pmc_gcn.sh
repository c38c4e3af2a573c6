#!/bin/bash
# PMC passes for every kernel of tools/gbench.py (graph conv + BN kernels of blocks 1 and 2): MFMA busy + clock,
# FETCH_SIZE, WRITE_SIZE, each in its own run.  Summaries: gpurun_out/pmc_hbm*_*.txt
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  local name=$1; local ctr=$2; shift 2
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d gpurun_out/pmc_$name -o p -- "$@" > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; exit 1; }
  DB=$(ls gpurun_out/pmc_$name/*.db gpurun_out/pmc_$name/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" > gpurun_out/pmc_$name.txt
  rm -rf gpurun_out/pmc_$name
}
rm -f gpurun_out/pmc_*.txt
for b in 1 2; do
  run hbm${b}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/gbench.py --block $b --reps 1
  run hbm${b}_fetch "FETCH_SIZE" python3 tools/gbench.py --block $b --reps 1
  run hbm${b}_write "WRITE_SIZE" python3 tools/gbench.py --block $b --reps 1
done
ls gpurun_out/pmc_*.txt
