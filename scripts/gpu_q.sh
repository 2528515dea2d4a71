cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
echo "--- default (fused FFN, rows >= 16384)"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['ffn_ms_per_step'], r['roofline']['launches_per_step'], r['roofline']['fused_fwd_kernel'] and r['roofline']['fused_fwd_kernel']['largest_launch'])"
echo "--- unfused"; DSVG_FFN_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['ffn_ms_per_step'], r['roofline']['launches_per_step'])"
done
echo "--- fused everywhere (min rows 0)"; DSVG_FFN_MIN_ROWS=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['ffn_ms_per_step'], r['roofline']['launches_per_step'])"
