#!/bin/bash
# ablations of the 4-frame forward form (diagnostic libs built with -DF16_ABL_*) and one PMC pass: clock + MFMA busy
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 1 2; do
  timeout -k 10 120 python -u tools/gbench.py --block $b --only "gcn_fwd " --reps 5
  for v in NOSTATS NOSTORE; do
    STEP_HIP_LIB=$PWD/step_amd/lib/libstep_abl_$v.so timeout -k 10 120 python -u tools/gbench.py --block $b --only "gcn_fwd " --reps 5 | sed "s/^/$v /"
  done
done
OUT=gpurun_out/pmc_fwd16f.txt
: > $OUT
for b in 1 2; do
  rm -rf gpurun_out/gp
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d gpurun_out/gp -o p -- python3 tools/gbench.py --block $b --only "gcn_fwd " --reps 1 > gpurun_out/gp.log 2>&1 || { tail -5 gpurun_out/gp.log; exit 1; }
  DB=$(ls gpurun_out/gp/*.db gpurun_out/gp/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" gcn_fwd | sed 's/(anonymous namespace):://g' | cut -c1-170 >> $OUT
  rm -rf gpurun_out/gp
done
cat $OUT
