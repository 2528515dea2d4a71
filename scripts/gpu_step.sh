#!/bin/bash
# whole GPU suite + default bench line (+ unfused A/B) + PMC of the fused FFN kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -n 8 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench_default.log | cut -c1-4000
DSVG_FFN_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32 > gpurun_out/bench_unfused.log 2>&1; echo "bench(unfused) rc=$?"; tail -n 1 gpurun_out/bench_unfused.log | cut -c1-1200
bash scripts/gpu_ffn_pmc.sh ffnpmc > /dev/null 2>&1; cat gpurun_out/ffnpmc_summary.txt | grep "ffn_\|gemm" | cut -c1-600
