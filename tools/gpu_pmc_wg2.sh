#!/bin/bash
# instruction mix / LDS counters of the fp32 temporal kernels (block-1 shape), one PMC pass
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_wg2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d gpurun_out/pmc_wg2 -o p -- python3 tools/kbench.py all --n 4096 --c 64 --stride ${S:-2} --t ${T:-240} --reps 1 > gpurun_out/pmc_wg2.log 2>&1 || { tail -5 gpurun_out/pmc_wg2.log; exit 1; }
DB=$(ls gpurun_out/pmc_wg2/*.db gpurun_out/pmc_wg2/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py pmc "$DB" tcn_ | grep -v "pack\|reduce" | sed 's/(anonymous namespace):://g' | cut -c1-150
rm -rf gpurun_out/pmc_wg2
