#!/bin/bash
# rocprofv3 kernel trace of the data-parallel graph step over a ONE-rank RCCL group (DSVG_FORCE_DDP=1): what the collectives and
# the eager optimiser tail cost next to the single-GPU graph step -> per-kernel CSV + per-launch timeline of the last step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 DSVG_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
NAME="${1:-prof_ddp}"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$NAME -o $NAME -- \
    python $GRAFT_REPO_ROOT/bench.py --gpus 1 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32 --no-torch-ref --graph 1 > $GRAFT_REPO_ROOT/gpurun_out/$NAME.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/$NAME -name "*.db" | head -1)
python scripts/rocpd_stats.py "$DB" gpurun_out/${NAME}_kernel_stats.csv > /dev/null
grep -i "nccl\|rccl\|adamw\|sumsq\|copy\|TOTAL\|fill\|at6native" gpurun_out/${NAME}_kernel_stats.csv | cut -c1-170
tail -1 gpurun_out/$NAME.log | cut -c1-300
rm -rf gpurun_out/$NAME
