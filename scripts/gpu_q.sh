cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "attn_" ) > gpurun_out/pytest_q.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/pytest_q.log | cut -c1-400 | tail -4
timeout 300 python scripts/attn_bench.py 2>&1 | tail -4
bash scripts/gpu_attn_pmc.sh 2>&1 | grep "BANK_CONFLICT\|^trace" | cut -c1-420
