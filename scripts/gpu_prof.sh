#!/bin/bash
# rocprofv3 kernel trace of the bf16 train step (packed encoder, eager) -> per-kernel CSV in gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
NAME="${1:-prof_bf16}"; shift
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$NAME -o $NAME -- \
    python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32 --no-torch-ref --graph 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$NAME.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/$NAME -name "*.db" | head -1)
python scripts/rocpd_stats.py "$DB" gpurun_out/${NAME}_kernel_stats.csv > /dev/null
head -45 gpurun_out/${NAME}_kernel_stats.csv | cut -c1-200; grep HIST gpurun_out/${NAME}_kernel_stats.csv
tail -1 gpurun_out/$NAME.log | cut -c1-400
rm -rf gpurun_out/$NAME
