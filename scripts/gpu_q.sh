cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])'
for v in 1 0 1 0 1 0; do echo "--- masked copy from ffn_bwd_dx: $v"; DSVG_FFN_BWD_MASKED=$v timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"; done
