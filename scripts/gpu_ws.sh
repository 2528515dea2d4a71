#!/bin/bash
# weight-stationary GEMM: parity tests, then the microbenchmark against the tiled LDS-DMA kernel (impl 4)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -x \
    -k "weight_stationary" ) > gpurun_out/ws_tests.log 2>&1
echo "rc=$?" >> gpurun_out/ws_tests.log
tail -n 12 gpurun_out/ws_tests.log
timeout 300 python scripts/gemm_bench.py --dtype bf16 --impls 4,5 --only "fwd QKV,FFN1" > gpurun_out/ws_bench.log 2>&1
cat gpurun_out/ws_bench.log | cut -c1-200
if [[ "$1" == "bench" ]]; then
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/ws_bench_step.log 2>&1; tail -n 3 gpurun_out/ws_bench_step.log | cut -c1-600
  DSVG_GEMM_WS=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/ws_bench_step_off.log 2>&1; tail -n 1 gpurun_out/ws_bench_step_off.log | cut -c1-300
fi
