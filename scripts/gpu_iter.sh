#!/bin/bash
# iteration loop: full GPU test-suite, bf16 train-step bench (default = packed / visible-first, eager), kernel profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --timeout 600 > gpurun_out/iter_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/iter_summary.log
tail -n 4 gpurun_out/iter_tests.log | cut -c1-300 >> gpurun_out/iter_summary.log
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/iter_bench.log 2>&1
tail -n 1 gpurun_out/iter_bench.log | cut -c1-1400 >> gpurun_out/iter_summary.log
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --graph 0 > gpurun_out/iter_bench_eager.log 2>&1
tail -n 1 gpurun_out/iter_bench_eager.log | cut -c1-400 >> gpurun_out/iter_summary.log
bash scripts/gpu_prof.sh "${1:-prof_iter}" > gpurun_out/iter_prof.log 2>&1
head -36 gpurun_out/iter_prof.log | cut -c1-140 >> gpurun_out/iter_summary.log
