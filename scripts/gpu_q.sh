cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -p no:cacheprovider -x -k "ffn or graph_replay or golden" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
DSVG_FFN_FUSED=0 timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
DSVG_FFN_BWD_FUSED=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
bash scripts/gpu_prof.sh prof_hybrid > /dev/null 2>&1; head -14 gpurun_out/prof_hybrid_kernel_stats.csv | cut -c1-150
