cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])'
run() { echo "--- $1"; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"; }
run default X=1
run "attention fused in the small stages too" DSVG_ATTN_MIN_ROWS=0
run "FFN fused in the small stages too" DSVG_FFN_MIN_ROWS=0
run "both" DSVG_ATTN_MIN_ROWS=0 DSVG_FFN_MIN_ROWS=0
run default X=1
