#!/bin/bash
# round 5, first contact of the role-specialised ffn_fwd: correctness + timing table, the existing FFN kernel tests (the debug
# stamp is a global store now: counted LDS waits in every FFN kernel), and a short default bench for the step time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 420 python scripts/ffn_rs_check.py ) > gpurun_out/ffn_rs_check.log 2>&1
echo "rs_check rc=$?"; cat gpurun_out/ffn_rs_check.log | cut -c1-900 | tail -n 90
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider --timeout 300 -x -k "ffn" ) > gpurun_out/ffn_tests.log 2>&1
echo "tests rc=$?"; tail -n 8 gpurun_out/ffn_tests.log | cut -c1-300
( time timeout 600 python bench.py --no-cpu-baseline --no-fp32 ) > gpurun_out/bench_short.log 2>&1
echo "bench rc=$?"; tail -n 3 gpurun_out/bench_short.log | cut -c1-1500
