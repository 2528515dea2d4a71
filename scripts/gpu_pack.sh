#!/bin/bash
# packed-encoder bring-up: new kernel tests, model parity, then the train-step bench packed vs padded.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --timeout 600 > gpurun_out/pack_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/pack_summary.log
tail -n 15 gpurun_out/pack_tests.log | cut -c1-300 >> gpurun_out/pack_summary.log
timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_packed.log 2>&1
tail -n 1 gpurun_out/bench_packed.log | cut -c1-2500 >> gpurun_out/pack_summary.log
DSVG_SKIP_INVISIBLE=0 timeout 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --pack-encoder 0 --no-cpu-baseline > gpurun_out/bench_padded.log 2>&1
tail -n 1 gpurun_out/bench_padded.log | cut -c1-2500 >> gpurun_out/pack_summary.log
timeout 300 python bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_fp32_packed.log 2>&1
tail -n 1 gpurun_out/bench_fp32_packed.log | cut -c1-2500 >> gpurun_out/pack_summary.log
