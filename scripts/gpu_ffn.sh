#!/bin/bash
# fused FFN: kernel tests + microbenchmark (+ timing probes with fewer hidden chunks: DSVG_FFN_DBG_CHUNKS)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider --timeout 300 -x -k "ffn" ) > gpurun_out/ffn_tests.log 2>&1
echo "tests rc=$?"; tail -n 25 gpurun_out/ffn_tests.log | cut -c1-300
timeout 600 python scripts/ffn_bench.py > gpurun_out/ffn_bench.log 2>&1
echo "bench rc=$?"; cat gpurun_out/ffn_bench.log | cut -c1-700
for n in 999; do
  echo "--- DSVG_FFN_DBG_CHUNKS=$n (timing probe, results invalid)"
  DSVG_FFN_DBG_CHUNKS=$n timeout 600 python scripts/ffn_bench.py --quick 2>&1 | grep rows | cut -c1-400
done
