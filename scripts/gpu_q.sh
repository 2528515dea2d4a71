cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -k "expand_rows or golden or slot_range or graph_replay or benchmark_size" ) > gpurun_out/pytest_q.log 2>&1; grep -E "passed|failed|Error|error|^E " gpurun_out/pytest_q.log | cut -c1-300 | tail -6
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])'
for v in 1 0 1 0; do echo "--- head dX by expand_rows: $v"; DSVG_HEAD_EXPAND=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"; done
