cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])'
run() { echo "--- $1"; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"; }
run default X=1
run "weight-stationary kernel for M <= 8192, K = 256 and 512" DSVG_GEMM_WS=2 DSVG_GEMM_WS_MIN_M=1 DSVG_GEMM_WS_MAX_M=8192
run "weight-stationary kernel for M <= 8192, K = 256 only" DSVG_GEMM_WS=1 DSVG_GEMM_WS_MIN_M=1 DSVG_GEMM_WS_MAX_M=8192
run default X=1
run "weight-stationary kernel for M <= 8192, K = 256 and 512" DSVG_GEMM_WS=2 DSVG_GEMM_WS_MIN_M=1 DSVG_GEMM_WS_MAX_M=8192
