#!/bin/bash
# LDS-DMA GEMM bring-up: kernel tests, side-by-side microbench of the three bf16 kernels, then the train-step bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "gemm" > gpurun_out/glds_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/glds_summary.log
tail -n 5 gpurun_out/glds_tests.log >> gpurun_out/glds_summary.log
timeout 300 python scripts/gemm_bench.py --dtype bf16 --impls 2,4,3 --vendor 1 > gpurun_out/gemm_bench_glds.log 2>&1
cat gpurun_out/gemm_bench_glds.log >> gpurun_out/glds_summary.log
timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench_glds1.log 2>&1
tail -n 2 gpurun_out/bench_glds1.log | cut -c1-1500 >> gpurun_out/glds_summary.log
DSVG_GEMM_STAGES=2 timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench_glds2.log 2>&1
tail -n 2 gpurun_out/bench_glds2.log | cut -c1-1500 >> gpurun_out/glds_summary.log
timeout 300 python bench.py --dtype fp32 --steps 5 --warmup 2 --graph 0 --no-cpu-baseline > gpurun_out/bench_fp32.log 2>&1
tail -n 1 gpurun_out/bench_fp32.log | cut -c1-1500 >> gpurun_out/glds_summary.log
