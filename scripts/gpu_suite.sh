#!/bin/bash
# the whole GPU test-suite, log under gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1
echo "rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 30 gpurun_out/pytest_gpu.log
