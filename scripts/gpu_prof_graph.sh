#!/bin/bash
# rocprofv3 kernel trace of the bf16 train step as it is TIMED (hipGraph replay) -> per-kernel CSV + per-launch timeline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
NAME="${1:-prof_graph}"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$NAME -o $NAME -- \
    python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32 --no-torch-ref --no-extra-legs --graph 1 > $GRAFT_REPO_ROOT/gpurun_out/$NAME.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/$NAME -name "*.db" | head -1)
python scripts/rocpd_stats.py "$DB" gpurun_out/${NAME}_kernel_stats.csv > /dev/null
head -30 gpurun_out/${NAME}_kernel_stats.csv | cut -c1-160; grep "HIST\|TOTAL" gpurun_out/${NAME}_kernel_stats.csv
tail -1 gpurun_out/$NAME.log | cut -c1-300
rm -rf gpurun_out/$NAME
