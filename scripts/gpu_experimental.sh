#!/bin/bash
# First contact of the kernels written without a GPU at the end of round 3 (opt-in in the library, tests skipped by default):
#   1. their gated tests (DSVG_EXPERIMENTAL=1: ffn_fwd stages = 5, attn_block_fwd with 4 ring slots, ffn_bwd_one with 4 and 3
#      ring slots), each under its own timeout - a kernel that hangs must not take the box along;
#   2. for the kernels whose tests passed: launch times against what they replace, the phase probe of the pipelined ffn_fwd;
#   3. the train step with every passing variant switched on against the default, same box (scripts/ab.sh).
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_experimental.sh'   -> gpurun_out/experimental_*.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time DSVG_EXPERIMENTAL=1 timeout 300 python -m pytest tests -q -p no:cacheprovider --timeout 120 -m gpu -k "pipelined" ) \
    > gpurun_out/experimental_tests.log 2>&1
rc=$?
echo "experimental tests (ffn_fwd stages 5) rc=$rc"; tail -n 30 gpurun_out/experimental_tests.log | cut -c1-300
( time DSVG_EXPERIMENTAL=1 timeout 300 python -m pytest tests -q -p no:cacheprovider --timeout 120 -m gpu -k "four_slot" ) \
    > gpurun_out/experimental_tests_attn.log 2>&1
rca=$?
echo "experimental tests (attn_block_fwd 4 slots) rc=$rca"; tail -n 30 gpurun_out/experimental_tests_attn.log | cut -c1-300
( time DSVG_EXPERIMENTAL=1 timeout 300 python -m pytest tests -q -p no:cacheprovider --timeout 120 -m gpu -k "one_launch" ) \
    > gpurun_out/experimental_tests_bwd_one.log 2>&1
rcb=$?
echo "experimental tests (ffn_bwd_one, 4 ring slots = all 160 KiB of LDS) rc=$rcb"; tail -n 30 gpurun_out/experimental_tests_bwd_one.log | cut -c1-300
( time DSVG_FFN_BWD_ONE_SLOTS=3 DSVG_EXPERIMENTAL=1 timeout 300 python -m pytest tests -q -p no:cacheprovider --timeout 120 -m gpu -k "one_launch" ) \
    > gpurun_out/experimental_tests_bwd_one_3slots.log 2>&1
rcb3=$?
echo "experimental tests (ffn_bwd_one, 3 ring slots) rc=$rcb3"; tail -n 12 gpurun_out/experimental_tests_bwd_one_3slots.log | cut -c1-300
for sl in 4 3; do
  { [ $sl -eq 4 ] && [ $rcb -ne 0 ]; } && continue
  { [ $sl -eq 3 ] && [ $rcb3 -ne 0 ]; } && continue
  echo "--- ffn_bwd_one, $sl ring slots"
  DSVG_FFN_BWD_ONE_SLOTS=$sl timeout 300 python scripts/ffn_bwd_one_probe.py 2>&1 | tee -a gpurun_out/experimental_ffn_bwd_one.log | cut -c1-300
done
for st in 3 4; do
  { [ $st -eq 4 ] && [ $rca -ne 0 ]; } && continue
  echo "--- attn_block_fwd, DSVG_ATTN_STAGES=$st"
  DSVG_ATTN_STAGES=$st timeout 300 python scripts/attn_bench.py 2>&1 | tee -a gpurun_out/experimental_attn_bench.log | cut -c1-300
done
if [ $rc -eq 0 ]; then
  timeout 300 python scripts/ffn_variant_probe.py 5 > gpurun_out/experimental_ffn_variants.log 2>&1
  echo "variant probe rc=$?"; cat gpurun_out/experimental_ffn_variants.log | cut -c1-400
  for fl in 1 2 3; do     # 1: no stage offset between the two waves of a SIMD, 2: s_setprio 1 for waves 4-7, 3: both
    echo "--- DSVG_FFN_PIPE_FLAGS=$fl"
    DSVG_FFN_PIPE_FLAGS=$fl timeout 300 python scripts/ffn_variant_probe.py 5 2>&1 | grep -E "rows +(63488|126976)" | tee -a gpurun_out/experimental_ffn_variants.log | cut -c1-400
  done
  for st in 4 5; do
    echo "--- phase probe, stages $st"
    PROBE_STAGES=$st timeout 200 python scripts/ffn_phase_probe.py 2>&1 | tee -a gpurun_out/experimental_ffn_phase.log | cut -c1-300
  done
fi
( time DSVG_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_model_gpu.py -q -p no:cacheprovider --timeout 300 -m gpu -k "experimental_kernels" ) \
    > gpurun_out/experimental_tests_model.log 2>&1
echo "experimental kernels inside the train step rc=$?"; tail -n 20 gpurun_out/experimental_tests_model.log | cut -c1-300
cfgs=("DSVG_FFN_STAGES=0")
[ $rc -eq 0 ] && cfgs+=("DSVG_FFN_STAGES=5" "DSVG_FFN_STAGES=5 DSVG_FFN_PIPE_FLAGS=2")
[ $rca -eq 0 ] && cfgs+=("DSVG_ATTN_STAGES=4")
[ $rcb -eq 0 ] && cfgs+=("DSVG_FFN_BWD_ONE=1")
[ $rcb -ne 0 ] && [ $rcb3 -eq 0 ] && cfgs+=("DSVG_FFN_BWD_ONE=1 DSVG_FFN_BWD_ONE_SLOTS=3")
[ $rc -eq 0 ] && [ $rca -eq 0 ] && [ $rcb -eq 0 ] && cfgs+=("DSVG_FFN_STAGES=5 DSVG_ATTN_STAGES=4 DSVG_FFN_BWD_ONE=1")
if [ ${#cfgs[@]} -gt 1 ]; then
  bash scripts/ab.sh "${cfgs[@]}" 2>&1 | tee gpurun_out/experimental_ab.log
fi
