#!/bin/bash
# same-box A/B of the frequency-domain weight gradient: operands shared through LDS vs every wave loading its own
# (STEP_DFT_WGRAD_DIRECT=1), at the three block shapes of the model
set -e
for args in "--c 64 --stride 2" "--c 64 --stride 1 --t 120" "--c 32 --stride 1"; do
  for i in 1 2; do
    echo "== lds:    $args"; timeout -k 10 120 python -u tools/dft_bench.py --no-direct --reps 10 --only wgrad $args | tail -n +2
    echo "== direct: $args"; STEP_DFT_WGRAD_DIRECT=1 timeout -k 10 120 python -u tools/dft_bench.py --no-direct --reps 10 --only wgrad $args | tail -n +2
  done
done
