#!/bin/bash
# PMC passes (each its own run): MFMA busy + clock, FETCH_SIZE, WRITE_SIZE for the wide temporal kernels and the
# graph-conv kernels.  Summaries go to gpurun_out/pmc_*.txt
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name counters -- program args
  local name=$1; local ctr=$2; shift 2
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d gpurun_out/pmc_$name -o p -- "$@" > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; exit 1; }
  DB=$(ls gpurun_out/pmc_$name/*.db gpurun_out/pmc_$name/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" > gpurun_out/pmc_$name.txt
  rm -rf gpurun_out/pmc_$name
}
for c in 128 256; do
  t=$((c == 128 ? 60 : 30))
  run wide${c}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/kbench.py all --n 2048 --c $c --stride 1 --t $t --reps 1
  run wide${c}_fetch "FETCH_SIZE" python3 tools/kbench.py all --n 2048 --c $c --stride 1 --t $t --reps 1
  run wide${c}_write "WRITE_SIZE" python3 tools/kbench.py all --n 2048 --c $c --stride 1 --t $t --reps 1
done
for b in 1 2; do
  run gcn${b}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/gbench.py --block $b --reps 1 --only gcn
  run gcn${b}_fetch "FETCH_SIZE" python3 tools/gbench.py --block $b --reps 1 --only gcn
  run gcn${b}_write "WRITE_SIZE" python3 tools/gbench.py --block $b --reps 1 --only gcn
done
grep -h "tcn_\|gcn_" gpurun_out/pmc_*.txt | cut -c1-160
