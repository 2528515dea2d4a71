cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -k "rccl" ) 2>&1 | grep -E "passed|failed" | tail -2
timeout 400 python scripts/ddp_probe.py 2>&1 | grep "force_ddp\|alone" | head -8
