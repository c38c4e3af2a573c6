#!/bin/bash
# same-box A/B of the spectrum's row granularity: 2 NF = round_up(T + P + 1, 4) vs the first version's 32
set -e
run() { timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-kernels --no-infer --no-bf16 | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'seq/s')"; }
for i in 1 2; do
  run rows4
  STEP_DFT_ROWS32=1 run rows32
done
