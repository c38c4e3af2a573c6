#!/bin/bash
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
b=${BLOCK:-2}
rm -rf gpurun_out/gbp$b
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/gbp$b -o p -- python3 tools/gbench.py --block $b --only gcn_bwd --reps 1 > gpurun_out/gbp$b.log 2>&1 || { tail -5 gpurun_out/gbp$b.log; exit 1; }
DB=$(ls gpurun_out/gbp$b/*.db gpurun_out/gbp$b/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py pmc "$DB" 16_kernel | sed 's/(anonymous namespace):://g' | cut -c1-140
rm -rf gpurun_out/gbp$b
