#!/bin/bash
# One GPU-box pass: GPU tests, smoke, bench, rocprofv3 kernel stats of the bench.  Every GPU step under its own limit.
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 420 python -u bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-kernels --no-bf16 --no-infer > gpurun_out/bench_rocprof.json 2> gpurun_out/rocprof.err || { tail -20 gpurun_out/rocprof.err; exit 1; }
DB=$(ls gpurun_out/prof/*.db gpurun_out/prof/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py stats "$DB" gpurun_out/bench_kernel_stats.csv
