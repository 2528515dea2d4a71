#!/bin/bash
# HBM traffic of the FFN GEMM launches of one train step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# `bench.py --ffn-replay`, summarised into profiles/ffn_traffic.json (read back by bench.py -> roofline.traffic).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
REPLAY=4
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ffn_$C -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --graph 0 --steps 2 --warmup 1 --no-cpu-baseline --ffn-replay $REPLAY \
      > $GRAFT_REPO_ROOT/gpurun_out/ffn_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_ffn_traffic.py $REPLAY gpurun_out/ffn_FETCH_SIZE gpurun_out/ffn_WRITE_SIZE gpurun_out/ffn_FETCH_SIZE.log \
    > gpurun_out/ffn_traffic.json
cat gpurun_out/ffn_traffic.json
