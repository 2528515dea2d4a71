cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/pytest_gpu.log | cut -c1-300 | tail -6
