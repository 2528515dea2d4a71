cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "rccl or attn_ or graph_replay" ) > gpurun_out/pytest_q.log 2>&1; tail -12 gpurun_out/pytest_q.log | cut -c1-250
timeout 300 python scripts/attn_bench.py 2>&1 | tail -4
echo "--- torchrun 1 rank (RCCL), graph split"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --no-cpu-baseline --no-fp32 --no-roofline 2>&1 | grep '^{"metric"\|calibration\|Error\|error' | cut -c1-400
