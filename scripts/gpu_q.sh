cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-fp32 2>&1 | grep '^{"metric"' | python -c '
import sys,json
r=json.loads(sys.stdin.read()); rf=r["roofline"]
print(r["ms_per_step"], r["value"])
print({k:rf[k] for k in ("frac","ffn_ms_per_step","launches_per_step")}, rf["matrix_launches_only"])
print(rf["fused_fwd_kernel"]["frac"], rf["fused_fwd_kernel"]["largest_launch"])
print(rf["fused_attn_fwd_kernel"]["ms_per_step"], rf["fused_attn_fwd_kernel"]["frac"], rf["fused_attn_fwd_kernel"]["largest_launch"])
print(rf["weight_grad_gemms"])'
