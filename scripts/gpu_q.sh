cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["ffn_ms_per_step"])'
for i in 1 2 3; do
echo "--- dW behind producers"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | grep '^{"metric"' | python -c "$show"
echo "--- dW at the end"; DSVG_FFN_BWD_ORDER=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | grep '^{"metric"' | python -c "$show"
done
