#!/bin/bash
# same-box A/B against a checkout of an older commit in _old/ (git archive <sha> step_amd tools/dft_bench.py include | tar -x -C _old/; make -C _old/step_amd/csrc)
set -e
for args in "$@"; do
  for i in 1 2; do
    echo "== old: $args"; timeout -k 10 120 python -u _old/tools/dft_bench.py --no-direct --reps 10 $args | tail -n +2
    echo "== new: $args"; timeout -k 10 120 python -u tools/dft_bench.py --no-direct --reps 10 $args | tail -n +2
  done
done
