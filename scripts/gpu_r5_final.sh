#!/bin/bash
# round 5, evidence on the final tree: the driver's bench command, the replayed-graph kernel trace + timeline, per-kernel PMC
# summaries (traffic / LDS conflicts / VALU per MFMA; clock + MFMA busy), the one-rank RCCL step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( time timeout 900 python bench.py ) > gpurun_out/r05_bench_default.log 2>&1
echo "bench rc=$?"; grep '^{' gpurun_out/r05_bench_default.log | cut -c1-400
bash scripts/gpu_prof_graph.sh r05_graph > gpurun_out/r05_prof_graph.txt 2>&1; tail -3 gpurun_out/r05_prof_graph.txt | cut -c1-200
bash scripts/gpu_step_pmc.sh r05_step_pmc > gpurun_out/r05_step_pmc.txt 2>&1; head -5 gpurun_out/r05_step_pmc_summary.txt | cut -c1-200
bash scripts/gpu_clock.sh > gpurun_out/r05_clock.txt 2>&1; head -8 gpurun_out/clock_pmc_summary.csv | cut -c1-200
( time timeout 600 env DSVG_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 1 --no-cpu-baseline --no-fp32 --no-roofline ) > gpurun_out/r05_bench_rccl_one_rank.log 2>&1
grep '^{' gpurun_out/r05_bench_rccl_one_rank.log | cut -c1-200
# net effect of the round's step-level changes on this box (all six knobs back to the round-4 behaviour vs the defaults)
bash scripts/ab.sh "DSVG_PACK_ONE=0 DSVG_DEFER_MORE=0 DSVG_GS_BWD_DG=0 DSVG_LN_BWD_MASKED=0 DSVG_STACK_GROUP=0 DSVG_HEAD_KPAD=0" "DSVG_PACK_ONE=1" > gpurun_out/r05_ab_round5_launch_changes.log 2>&1
cat gpurun_out/r05_ab_round5_launch_changes.log
