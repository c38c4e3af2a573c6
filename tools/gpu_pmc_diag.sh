#!/bin/bash
# Diagnostic PMC passes (each its own run) on one stage of the frequency-domain conv: ONLY="fwd: transform" ARGS="--c 64 ..."
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
ONLY=${ONLY:-fwd: transform}
KERN=${KERN:-dft_gemm}
run() {
  local name=$1; local ctr=$2
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 90 rocprofv3 --pmc $ctr -d gpurun_out/pmc_$name -o p -- python3 tools/dft_bench.py --reps 1 --no-direct --only "$ONLY" $ARGS > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; exit 1; }
  DB=$(ls gpurun_out/pmc_$name/*.db gpurun_out/pmc_$name/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" $KERN > gpurun_out/pmc_$name.txt
  rm -rf gpurun_out/pmc_$name
  cat gpurun_out/pmc_$name.txt | cut -c1-220
}
run diag_a "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
run diag_b "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run diag_c "FETCH_SIZE"
run diag_d "WRITE_SIZE"
