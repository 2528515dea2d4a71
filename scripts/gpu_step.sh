#!/bin/bash
# whole GPU suite + default bench line (+ unfused A/B)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_fused.log 2>&1; echo "bench rc=$?"; tail -n 2 gpurun_out/bench_fused.log | cut -c1-1500
DSVG_FFN_FUSED=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_unfused.log 2>&1; echo "bench(unfused) rc=$?"; tail -n 1 gpurun_out/bench_unfused.log | cut -c1-600
