#!/bin/bash
# same-box A/B of two builds of the library: step_amd/lib/ab_base.so (built from the commit to compare against) vs the
# in-tree libstep_hip.so, interleaved.  usage: tools/gpu_ab_lib.sh "<dft_bench arguments>" ["<more arguments>" ...]
set -e
for args in "$@"; do
  for i in 1 2; do
    echo "== base: $args"
    STEP_HIP_LIB=$PWD/step_amd/lib/ab_base.so timeout -k 10 120 python -u tools/dft_bench.py --no-direct --reps 10 $args | tail -n +2
    echo "== new:  $args"
    timeout -k 10 120 python -u tools/dft_bench.py --no-direct --reps 10 $args | tail -n +2
  done
done
