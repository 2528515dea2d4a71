#!/bin/bash
# Round-3 development call: new-kernel parity tests first, then the regression suite, the bench line and a kernel trace of
# the graph step.  Everything under `timeout`; logs under gpurun_out/ (merged back by gpurun).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
WHAT="${1:-gs,suite,bench,prof}"
: > gpurun_out/summary.log
say() { echo "$@" | tee -a gpurun_out/summary.log; }
if [[ "$WHAT" == *gs* ]]; then
  ( time timeout 900 python -m pytest tests/test_group_stage_gpu.py -q -p no:cacheprovider --timeout 300 -x --no-header -rN ) > gpurun_out/gs_tests.log 2>&1
  GS_RC=$?
  say "=== gs tests rc=$GS_RC"; grep -E "passed|failed|error" gpurun_out/gs_tests.log | tail -3 | tee -a gpurun_out/summary.log
  if [[ $GS_RC -ne 0 ]]; then
    # not stopping at the first failure: every case, so that one call shows the whole picture
    ( timeout 900 python -m pytest tests/test_group_stage_gpu.py -q -p no:cacheprovider --timeout 300 --no-header ) > gpurun_out/gs_tests_all.log 2>&1
    grep -E "^(FAILED|ERROR)|layer [01] |forward |backward |passed|failed" gpurun_out/gs_tests_all.log | cut -c1-260 | head -150 >> gpurun_out/summary.log
    export DSVG_GS_FUSED=0
    say "(group-stage kernels OFF for the rest of this call)"
  fi
fi
if [[ "$WHAT" == *suite* ]]; then
  ( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 --no-header --deselect tests/test_group_stage_gpu.py ) > gpurun_out/pytest_gpu.log 2>&1
  say "=== gpu suite rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-300 | tail -25 | tee -a gpurun_out/summary.log
fi
if [[ "$WHAT" == *bench* ]]; then
  ( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
  say "=== bench rc=$?"; grep -E "^\[bench|Error|error" gpurun_out/bench_default.log | cut -c1-400 | tail -14 | tee -a gpurun_out/summary.log
  grep -E '^\{"metric' gpurun_out/bench_default.log | cut -c1-1200 | tee -a gpurun_out/summary.log
  if [[ -z "$DSVG_GS_FUSED" ]]; then
    ( timeout 600 env DSVG_GS_FUSED=0 python bench.py --no-cpu-baseline --no-fp32 --no-torch-ref ) > gpurun_out/bench_gs_off.log 2>&1
    say "=== bench (group-stage kernels off) rc=$?"; grep -E '^\{"metric' gpurun_out/bench_gs_off.log | cut -c1-400 | tee -a gpurun_out/summary.log
  fi
fi
if [[ "$WHAT" == *prof* ]]; then
  bash scripts/gpu_prof_graph.sh prof_graph > gpurun_out/prof_graph.txt 2>&1
  say "=== graph-step kernel trace"; head -24 gpurun_out/prof_graph.txt | cut -c1-150 >> gpurun_out/summary.log; grep "HIST\|TOTAL" gpurun_out/prof_graph.txt >> gpurun_out/summary.log
fi
say "=== done"
