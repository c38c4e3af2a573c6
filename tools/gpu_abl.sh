#!/bin/bash
# timing of diagnostic library variants (step_amd/lib/libstep_abl_*.so) against the product library, forward graph conv
set -e
for b in 1 2; do
  timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn_fwd_stats --reps 10
  for lib in step_amd/lib/libstep_abl_*.so; do
    STEP_HIP_LIB=$PWD/$lib timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn_fwd_stats --reps 10 | sed "s/^/$(basename $lib .so | sed s/libstep_abl_//) /"
  done
  timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn_fwd_stats --reps 10
done
