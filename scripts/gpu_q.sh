cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x ) > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error|error|^E " gpurun_out/pytest_gpu.log | cut -c1-300 | tail -8
python scripts/attn_drop_probe.py 2>&1 | tail -2
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])'
for i in 1 2; do
echo "--- group-stage attention on the tiled MFMA kernels"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"
echo "--- VALU kernels (before)"; DSVG_ATTN_MFMA_MIN_S=17 timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"
done
