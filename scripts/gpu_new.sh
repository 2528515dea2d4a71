#!/bin/bash
# GPU check of the newest rows only (batch assembly, label conditioning, hierarch path) + a micro-timing of the
# assembly kernel; logs under gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 600 python -m pytest tests/test_dataset.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 \
    -k "dataset or assembled or variant or full_size_batch or fonts or hierarch or label" ) > gpurun_out/new_tests.log 2>&1
echo "rc=$?" >> gpurun_out/new_tests.log
tail -n 15 gpurun_out/new_tests.log
timeout 300 python scripts/assemble_bench.py > gpurun_out/assemble_bench.log 2>&1
tail -n 6 gpurun_out/assemble_bench.log
