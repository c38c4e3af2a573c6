#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gcn.log 2>&1 || { tail -30 gpurun_out/pytest_gcn.log; exit 1; }
tail -2 gpurun_out/pytest_gcn.log
for b in 1 2; do
  echo "== block $b new"; timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn_fwd
  echo "== block $b old"; STEP_GCN_FWD_V1=1 timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn_fwd
done
