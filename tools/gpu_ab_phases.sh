#!/bin/bash
# same-box A/B of the stride-2 temporal conv: two stride-1 convs on the even / odd frames (dft.phases) vs the
# strided-matrix form (STEP_DFT_NO_PHASES=1): stage timings, then the whole fp32 train step, interleaved
set -e
mkdir -p gpurun_out
echo "== phases"; timeout -k 10 200 python -u tools/dft_bench.py --no-direct --reps 10
echo "== strided matrices"; STEP_DFT_NO_PHASES=1 timeout -k 10 200 python -u tools/dft_bench.py --no-direct --reps 10
run() { timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-kernels --no-infer --no-bf16 | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'seq/s')"; }
for i in 1 2; do
  run phases
  STEP_DFT_NO_PHASES=1 run strided
done
