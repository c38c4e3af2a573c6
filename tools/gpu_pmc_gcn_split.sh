#!/bin/bash
# PMC passes behind DESIGN 4.2c's finding on the graph-conv data-gradient kernels (block 2 shape, N = 4096, split twin):
# one counter group per run.  Output: gpurun_out/pmc_gcn_split.txt
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gcn_split.txt
echo "# rocprofv3 --pmc <group> -- python3 tools/gbench.py --block 2 --only gcn_bwd_split --reps 1   (one group per run; mean over the 2 launches)" > $OUT
pass() {
  rm -rf gpurun_out/gp
  timeout -s KILL 120 rocprofv3 --pmc $1 -d gpurun_out/gp -o p -- python3 tools/gbench.py --block 2 --only gcn_bwd_split --reps 1 > gpurun_out/gp.log 2>&1 || { tail -5 gpurun_out/gp.log; exit 1; }
  DB=$(ls gpurun_out/gp/*.db gpurun_out/gp/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" gcn_bwd_ | grep -v finalize | sed 's/(anonymous namespace):://g' | cut -c1-170 >> $OUT
  rm -rf gpurun_out/gp
}
pass "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
pass "FETCH_SIZE"
pass "WRITE_SIZE"
pass "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
pass "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"
pass "TCC_HIT_sum TCC_MISS_sum"
pass "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum"
cat $OUT
