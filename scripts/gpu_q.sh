cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_gpu.py -q -p no:cacheprovider -x -s 2>&1 | grep "C4\|C5\|passed\|failed\|Error" | cut -c1-300
timeout 600 python scripts/secondary_bench.py 2>&1 | tee gpurun_out/secondary_bench.log | cut -c1-250
