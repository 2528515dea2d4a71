cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python scripts/attn_bench.py 2>&1 | tail -8
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -x ) > gpurun_out/pytest_q.log 2>&1; tail -5 gpurun_out/pytest_q.log | cut -c1-220
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["ffn_ms_per_step"])'
for i in 1 2; do
echo "--- fused attention"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | grep '^{"metric"' | python -c "$show"
echo "--- unfused attention"; DSVG_ATTN_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | grep '^{"metric"' | python -c "$show"
done
