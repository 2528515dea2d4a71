cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error|error|^E " gpurun_out/pytest_gpu.log | cut -c1-300 | tail -6
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])'
for v in 256 512 192 384 256; do
echo "--- 4-stage GEMM for launches of <= $v workgroups"; DSVG_GEMM_DEEP_WGS=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"
done
