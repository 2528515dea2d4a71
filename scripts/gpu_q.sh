cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "deferred or graph_replay or rccl" ) > gpurun_out/pytest_q.log 2>&1; tail -5 gpurun_out/pytest_q.log | cut -c1-250
show='import sys,json; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"], r["config"].get("hip_graph"))'
for i in 1 2; do
echo "--- side stream for small dW"; DSVG_SIDE_STREAM=1 timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"
echo "--- single stream"; timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"
done
echo "--- side stream, all dW"; DSVG_SIDE_STREAM=1 DSVG_SIDE_MAX_ROWS=100000000 timeout 600 python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"' | python -c "$show"
echo "--- 1-rank RCCL, graph split"; DSVG_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"\|calibration' | cut -c1-300 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], r['config'].get('hip_graph'))
    else: print(l.strip())"
