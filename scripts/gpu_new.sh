#!/bin/bash
# GPU check of the newest rows only; logs under gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 \
    -k "${1:-match or selfmatch or self_matching}" ) > gpurun_out/new_tests.log 2>&1
echo "rc=$?" >> gpurun_out/new_tests.log
tail -n 25 gpurun_out/new_tests.log
