cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3 4; do
( DSVG_FORCE_DDP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2971$i bench.py --gpus 1 --no-cpu-baseline --no-fp32 --no-roofline ) > gpurun_out/rccl_$i.log 2>&1; echo "run $i rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"hip_graph": [a-z]*\|capture failed' gpurun_out/rccl_$i.log | tr '\n' ' '; echo
done
( timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -k "rccl or graph_replay or self_matching or autoregressive_training" ) 2>&1 | grep -E "passed|failed" | tail -2
