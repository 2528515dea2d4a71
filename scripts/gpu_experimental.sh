#!/bin/bash
# First contact of the kernels written without a GPU at the end of round 3 (opt-in in the library, tests skipped by default):
#   1. their gated tests (DSVG_EXPERIMENTAL=1), each under its own timeout - a kernel that hangs must not take the box along;
#   2. bit-equality + launch times of the ffn_fwd variants, the phase probe of the pipelined one;
#   3. if (1) passed: the train step with the variant switched on against the default, same box (scripts/ab.sh).
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_experimental.sh'   -> gpurun_out/experimental_*.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time DSVG_EXPERIMENTAL=1 timeout 300 python -m pytest tests -q -p no:cacheprovider --timeout 120 -m gpu -k "pipelined" ) \
    > gpurun_out/experimental_tests.log 2>&1
rc=$?
echo "experimental tests rc=$rc"; tail -n 30 gpurun_out/experimental_tests.log | cut -c1-300
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
if [ $rc -eq 0 ]; then
  bash scripts/ab.sh "DSVG_FFN_STAGES=0" "DSVG_FFN_STAGES=5" "DSVG_FFN_STAGES=5 DSVG_FFN_PIPE_FLAGS=2" 2>&1 | tee gpurun_out/experimental_ab.log
fi
