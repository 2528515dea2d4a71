cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "attn_ or benchmark_size_512 or greedy_sample" -s ) > gpurun_out/pytest_q.log 2>&1; grep -E "passed|failed|N=512|Error|error" gpurun_out/pytest_q.log | cut -c1-400 | tail -12
timeout 600 python scripts/secondary_bench.py 2>&1 | grep "^C5" | cut -c1-200
