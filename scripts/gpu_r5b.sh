#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 300 python scripts/ffn_rs_probe.py ) > gpurun_out/ffn_rs_probe.log 2>&1
echo "probe rc=$?"; grep "cycles\|iteration 8" gpurun_out/ffn_rs_probe.log | cut -c1-600
for pr in $PROBES; do
  echo "--- DSVG_FFN_RS_PROBE=$pr (timing experiment, results invalid)"
  DSVG_FFN_RS_PROBE=$pr timeout 300 python scripts/ffn_rs_check.py --quick 2>&1 | grep "rows  " -A1 | cut -c1-600
done
