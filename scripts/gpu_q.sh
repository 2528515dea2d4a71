cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -k "slot_range" ) > gpurun_out/pytest_q.log 2>&1; grep -E "passed|failed|Error|error|^E " gpurun_out/pytest_q.log | cut -c1-300 | tail -8
