#!/bin/bash
# bf16-operand / split-operand temporal conv: tests, then forward / input-gradient / weight-gradient timings
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -x -q --timeout 120 --timeout-method thread -k "wgrad or bf16 or tcn or split" > gpurun_out/pytest_bf16.log 2>&1 || { tail -30 gpurun_out/pytest_bf16.log; exit 1; }
tail -1 gpurun_out/pytest_bf16.log
timeout -k 10 100 python -u tools/kbench.py all --n 4096 --c 64 --stride 2 --t 240 --reps 3 --bf16
timeout -k 10 100 python -u tools/kbench.py all --n 4096 --c 64 --stride 1 --t 120 --reps 3 --bf16
timeout -k 10 100 python -u tools/kbench.py all --n 4096 --c 32 --stride 1 --t 240 --reps 3 --bf16
timeout -k 10 100 python -u tools/kbench.py all --n 4096 --c 64 --stride 2 --t 240 --reps 3 --split
timeout -k 10 100 python -u tools/kbench.py all --n 4096 --c 64 --stride 1 --t 120 --reps 3 --split
timeout -k 10 100 python -u tools/kbench.py all --n 4096 --c 32 --stride 1 --t 240 --reps 3 --split
