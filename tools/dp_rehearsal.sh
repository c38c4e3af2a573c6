#!/bin/bash
# 2-rank rehearsal of bench.py's data-parallel path on a 1-GPU box (gloo, both ranks on cuda:0).
# The real multi-GPU run uses RCCL: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
export STEP_DIST_BACKEND=gloo STEP_LOCAL_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1
exec python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 2 --warmup 1 --batch 256 --no-cpu --no-kernels "$@"
