#!/bin/bash
# same-box A/B of the whole fp32 train step: round-3 forms (fused statistics, 4-frame graph-conv forward) vs the
# separate statistics passes and the older forward forms; each arm twice, interleaved
set -e
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-kernels --no-infer --no-bf16 $BENCH_EXTRA | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'seq/s')"; }
for i in 1 2; do
  run new
  STEP_SEPARATE_STATS=1 STEP_GCN_FWD_OLD=1 run old
  STEP_SEPARATE_STATS=1 run new_kernels_separate_stats
done
