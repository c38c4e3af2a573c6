#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/gcn_ab.py > gpurun_out/ab.log 2>&1 || { tail gpurun_out/ab.log; exit 1; }
tail -1 gpurun_out/ab.log
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gcn.log 2>&1 || { tail -30 gpurun_out/pytest_gcn.log; exit 1; }
tail -1 gpurun_out/pytest_gcn.log
for b in 0 1 2; do
  timeout -k 10 120 python -u tools/gbench.py --block $b --only gcn
done
