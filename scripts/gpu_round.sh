#!/bin/bash
# One gpurun call that mirrors the driver's round-end checks: smoke, the GPU test-suite, the default bench line (with
# roofline + fp32 + cpu_baseline), the unfused-FFN A/B, secondary workloads, train sanity and a kernel profile.
# Everything is wrapped in `timeout`; logs go to gpurun_out/ (merged back by gpurun).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
WHAT="${1:-all}"
run() { # name, timeout, cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.log
  ( time timeout "$t" "$@" ) > "gpurun_out/$name.log" 2>&1
  echo "rc=$? ($name)" | tee -a gpurun_out/summary.log
  tail -n 4 "gpurun_out/$name.log" | cut -c1-2500 | tee -a gpurun_out/summary.log
}
: > gpurun_out/summary.log
if [[ "$WHAT" == "all" || "$WHAT" == *smoke* ]]; then
  run smoke 600 python -c "import __graft_entry__ as g; g.smoke()"
fi
if [[ "$WHAT" == "all" || "$WHAT" == *test* ]]; then
  run pytest_gpu 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -s
fi
if [[ "$WHAT" == "all" || "$WHAT" == *bench* ]]; then
  run bench_default 900 python bench.py
  run bench_unfused_ffn 600 env DSVG_FFN_FUSED=0 python bench.py --no-cpu-baseline --no-fp32
  run bench_unfused_attn 600 env DSVG_ATTN_FUSED=0 python bench.py --no-cpu-baseline --no-fp32 --no-roofline
  run bench_immediate_reduce 600 env DSVG_DEFER_REDUCE=0 python bench.py --no-cpu-baseline --no-fp32 --no-roofline
  run bench_rccl_one_rank 600 env DSVG_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 1 --no-cpu-baseline --no-fp32 --no-roofline
  run bench_bf16_eager 600 python bench.py --graph 0 --no-cpu-baseline --no-fp32
  run bench_bf16_padded 600 env DSVG_SKIP_INVISIBLE=0 DSVG_COMPACT_HEAD=0 python bench.py --pack-encoder 0 --no-cpu-baseline --no-fp32
fi
if [[ "$WHAT" == "all" || "$WHAT" == *second* ]]; then
  run attn_microbench 300 python scripts/attn_bench.py
  run secondary_bench 600 python scripts/secondary_bench.py
  bash scripts/gpu_prof_graph.sh prof_graph > gpurun_out/prof_graph.txt 2>&1
  run train_sanity 600 python scripts/train_sanity.py
fi
if [[ "$WHAT" == "all" || "$WHAT" == *prof* ]]; then
  bash scripts/gpu_prof.sh prof_round > gpurun_out/prof_round.txt 2>&1
  head -40 gpurun_out/prof_round.txt | cut -c1-150 >> gpurun_out/summary.log
fi
echo "=== done" >> gpurun_out/summary.log
