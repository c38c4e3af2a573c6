#!/bin/bash
# is the small-batch bf16 step (configs[2]: per-GPU batch 512) bound by the GPU or by launches?  kernel-time sum vs wall
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_sb
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sb -o p -- python3 bench.py --batch 512 --mfma bf16 --steps 10 --warmup 2 --no-cpu --no-kernels --no-infer --no-bf16 > gpurun_out/sb.json 2> gpurun_out/sb.err || { tail -5 gpurun_out/sb.err; exit 1; }
cut -c1-160 gpurun_out/sb.json
DB=$(ls gpurun_out/prof_sb/*.db gpurun_out/prof_sb/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/sb_kernel_stats.csv > gpurun_out/sb_stats.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/sb_kernel_stats.csv')))
tot=sum(float(r['TotalDurationUs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("kernel time %.2f ms per step over 12 steps, %d launches per step" % (tot/12/1e3, calls/12))
small=[r for r in rows if float(r['AverageUs'])<15]
print("launches under 15 us: %d per step, %.2f ms per step" % (sum(int(r['Calls']) for r in small)/12, sum(float(r['TotalDurationUs']) for r in small)/12/1e3))
PY
rm -rf gpurun_out/prof_sb
