#!/bin/bash
# Data-parallel REHEARSAL on the one-GPU box: 2 ranks share the card, collectives over gloo through host memory
# (RCCL refuses two ranks on one device) - not a scaling number.  What it puts on record: the step time of the
# configs[2]-sized shard (per-GPU batch 512, bf16 MFMA operands) alone, as 2 ranks with the gradient bucket only,
# and as 2 ranks with synchronised BatchNorm (7 + 6 small fp64 collectives more per step).
set -e
mkdir -p gpurun_out
OUT=gpurun_out/dp_rehearsal.jsonl
: > $OUT
COMMON="--batch 512 --steps 10 --warmup 3 --no-cpu --no-kernels --no-infer --no-bf16 --mfma bf16"
timeout -k 10 300 python bench.py --gpus 1 $COMMON | grep "^{" >> $OUT
timeout -k 10 300 python bench.py --gpus 2 $COMMON | grep "^{" >> $OUT
timeout -k 10 300 python bench.py --gpus 2 $COMMON --sync-bn | grep "^{" >> $OUT
python - <<'PY'
import json
for l in open("gpurun_out/dp_rehearsal.jsonl"):
    d = json.loads(l)
    print(d["n_gpus"], d["config"]["parallelism"], d["ms_per_step"], "ms/step", d["value"], "seq/s", "rehearsal" if d.get("rehearsal") else "")
PY
