#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/bf16_parity.log
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider --timeout 300 -x -k "ffn" ) > gpurun_out/ffn_tests.log 2>&1
echo "tests rc=$?"; tail -n 12 gpurun_out/ffn_tests.log | cut -c1-400
( time timeout 420 python scripts/ffn_rs_check.py --timing-only ) > gpurun_out/ffn_rs_check.log 2>&1
echo "rs_check rc=$?"; grep "rows  \|^   " gpurun_out/ffn_rs_check.log | cut -c1-900
( time timeout 900 python -m pytest tests/test_model_gpu.py -q -p no:cacheprovider --timeout 600 -x -k "bf16_model_tracks or benchmark_size_512" ) > gpurun_out/bf16_tests.log 2>&1
echo "bf16 tests rc=$?"; tail -n 5 gpurun_out/bf16_tests.log | cut -c1-400
grep "direction" gpurun_out/bf16_parity.log | cut -c1-500
( time timeout 600 python bench.py --no-cpu-baseline --no-fp32 ) > gpurun_out/bench_short.log 2>&1
echo "bench rc=$?"; grep '^{' gpurun_out/bench_short.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['fused_fwd_kernel'], d['roofline']['fused_attn_fwd_kernel'])"
