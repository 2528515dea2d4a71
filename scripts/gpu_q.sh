cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x ) > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log | cut -c1-200
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["ffn_ms_per_step"], r["roofline"].get("matrix_launches_only"))'
for i in 1 2; do
echo "--- deferred reductions"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | grep '^{"metric"' | python -c "$show"
echo "--- immediate reductions"; DSVG_DEFER_REDUCE=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | grep '^{"metric"' | python -c "$show"
done
echo "--- deferred, eager"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 --graph 0 2>&1 | grep '^{"metric"' | python -c "$show"
