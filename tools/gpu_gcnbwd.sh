#!/bin/bash
# per-kernel times of the graph-conv backward (blocks 1 and 2) under rocprofv3, then an MFMA-busy PMC pass
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 1 2; do
  rm -rf gpurun_out/gb$b
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/gb$b -o p -- python3 tools/gbench.py --block $b --only gcn_bwd --reps 3 > gpurun_out/gb$b.log 2>&1 || { tail -5 gpurun_out/gb$b.log; exit 1; }
  DB=$(ls gpurun_out/gb$b/*.db gpurun_out/gb$b/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py stats "$DB" gpurun_out/gb${b}_stats.csv | grep -E "gcn|sum_rows" || true
  rm -rf gpurun_out/gb$b
done
if [ -n "$PMC" ]; then
for b in 1 2; do
  rm -rf gpurun_out/gbp$b
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/gbp$b -o p -- python3 tools/gbench.py --block $b --only gcn_bwd --reps 1 > gpurun_out/gbp$b.log 2>&1 || { tail -5 gpurun_out/gbp$b.log; exit 1; }
  DB=$(ls gpurun_out/gbp$b/*.db gpurun_out/gbp$b/*/*.db 2>/dev/null | head -1)
  python tools/rocpd_summary.py pmc "$DB" gcn_bwd | cut -c1-150
  rm -rf gpurun_out/gbp$b
done
fi
