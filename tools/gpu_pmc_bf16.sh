#!/bin/bash
# PMC passes over the bf16-operand temporal kernels (block-2 shape: C = 64, stride 1, T = 120), each in its own run
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
S=${S:-1}; T=${T:-120}; C=${C:-64}
run bf_a "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" python3 tools/kbench.py all --bf16 --n 4096 --c $C --stride $S --t $T --reps 1
cat gpurun_out/pmc_bf_a.txt | cut -c1-260
