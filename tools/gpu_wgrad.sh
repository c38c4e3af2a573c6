#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -x -q --timeout 120 --timeout-method thread -k "wgrad or bf16 or tcn" > gpurun_out/pytest_wgrad.log 2>&1 || { tail -30 gpurun_out/pytest_wgrad.log; exit 1; }
tail -1 gpurun_out/pytest_wgrad.log
for s in 1 2; do
  timeout -k 10 100 python -u tools/kbench.py wgrad --n 4096 --c 64 --stride $s --t 240 --reps 3
  timeout -k 10 100 python -u tools/kbench.py wgrad --n 4096 --c 64 --stride $s --t 240 --reps 3 --bf16
  timeout -k 10 100 python -u tools/kbench.py wgrad --n 4096 --c 64 --stride $s --t 240 --reps 3 --split
done
timeout -k 10 100 python -u tools/kbench.py wgrad --n 4096 --c 32 --stride 1 --t 240 --reps 3
timeout -k 10 100 python -u tools/kbench.py wgrad --n 4096 --c 32 --stride 1 --t 240 --reps 3 --bf16
timeout -k 10 100 python -u tools/kbench.py wgrad --n 4096 --c 32 --stride 1 --t 240 --reps 3 --split
