#!/bin/bash
# A/B of the round-3 4-frame forward graph-conv form: parity against the first form, op tests, timing old vs new
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/gcn_ab.py > gpurun_out/ab_f16.log 2>&1 || { tail -20 gpurun_out/ab_f16.log; exit 1; }
tail -1 gpurun_out/ab_f16.log
timeout -k 10 900 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_dist.py tests/test_gpu_bf16.py > gpurun_out/pytest_gcn.log 2>&1 || { tail -30 gpurun_out/pytest_gcn.log; exit 1; }
tail -2 gpurun_out/pytest_gcn.log
for b in 0 1 2; do
  timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn_fwd --reps 5
  STEP_GCN_FWD_OLD=1 timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn_fwd --reps 5 | sed 's/^/OLD /'
done
