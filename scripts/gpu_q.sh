cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "deferred or aliasing or graph_replay" ) > gpurun_out/pytest_q.log 2>&1; tail -3 gpurun_out/pytest_q.log | cut -c1-200
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["ffn_ms_per_step"])'
for b in 512 384 256 512 256; do
echo "--- split-K target blocks $b"; DSVG_SPLITK_BLOCKS=$b timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])'
done
