#!/bin/bash
# parity-gap tests of round 2 (N=512 fp32 vs oracle, greedy_sample goldens, bf16 tracking log, aliasing assert)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/bf16_parity.log
( time timeout 900 python -m pytest tests/test_model_gpu.py -q -p no:cacheprovider --timeout 600 -s \
   -k "benchmark_size_512 or greedy_sample_matches or bf16_model_tracks or aliasing" ) > gpurun_out/parity_tests.log 2>&1
echo "rc=$?"; tail -n 30 gpurun_out/parity_tests.log | cut -c1-400
cat gpurun_out/bf16_parity.log
