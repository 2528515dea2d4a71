cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "ffn" 2>&1 | tail -2
timeout 300 python scripts/ffn_bench.py --quick 2>&1 | grep rows | cut -c1-300
DSVG_FFN_STAGES=4 timeout 300 python scripts/ffn_bench.py --quick 2>&1 | grep rows | cut -c1-300
