#!/bin/bash
# quick GPU iteration: kernel tests + GEMM microbench + one bf16 bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" > gpurun_out/quick.log
tail -3 gpurun_out/pytest_gpu.log >> gpurun_out/quick.log
timeout 300 python scripts/gemm_bench.py --dtype bf16 > gpurun_out/gemm_bench_bf16.log 2>&1
cat gpurun_out/gemm_bench_bf16.log >> gpurun_out/quick.log
timeout 600 python bench.py --dtype bf16 --steps 10 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1
tail -2 gpurun_out/bench_bf16.log | cut -c1-500 >> gpurun_out/quick.log
