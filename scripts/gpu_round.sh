#!/bin/bash
# One gpurun call: smoke, GPU test-suite, benches and a rocprofv3 kernel summary.  Everything is wrapped in
# `timeout` so a hung kernel cannot hold the box; logs go to gpurun_out/ (merged back by gpurun).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export PYTHONDONTWRITEBYTECODE=1
WHAT="${1:-all}"

run() { # name, timeout, cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.log
  ( time timeout "$t" "$@" ) > "gpurun_out/$name.log" 2>&1
  echo "rc=$? ($name)" | tee -a gpurun_out/summary.log
  tail -n 6 "gpurun_out/$name.log" | cut -c1-400 | tee -a gpurun_out/summary.log
}

: > gpurun_out/summary.log
rocm-smi --showproductname 2>/dev/null | head -8 >> gpurun_out/summary.log
if [[ "$WHAT" == "all" || "$WHAT" == *smoke* ]]; then
  run smoke 600 python -c "import __graft_entry__ as g; g.smoke()"
fi
if [[ "$WHAT" == "all" || "$WHAT" == *test* ]]; then
  run pytest_gpu 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -rA
fi
if [[ "$WHAT" == "all" || "$WHAT" == *bench* ]]; then
  run bench_fp32 900 python bench.py --dtype fp32 --steps 5 --warmup 2 --graph 0
  run bench_bf16 900 python bench.py --dtype bf16 --steps 10 --warmup 3 --graph 0
  run bench_bf16_graph 900 python bench.py --dtype bf16 --steps 20 --warmup 5 --graph 1 --no-cpu-baseline
fi
if [[ "$WHAT" == "all" || "$WHAT" == *prof* ]]; then
  run rocprof_bf16 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bf16 -o bf16 -- \
      python bench.py --dtype bf16 --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline
  find gpurun_out/prof_bf16 -name "*kernel_stats*" | head -3 >> gpurun_out/summary.log
fi
echo "=== done" >> gpurun_out/summary.log
