#!/bin/bash
# round 6, evidence on the final tree: FFN traffic on the final kernel source, the driver's bench command, the replayed-graph kernel
# trace + timeline, the per-kernel PMC summary (traffic / LDS conflicts / VALU per MFMA), the net effect of the round's step-level
# change (weight images requested into L2 up front) on this box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 500 bash scripts/gpu_ffn_traffic.sh > gpurun_out/r06_ffn_traffic.txt 2>&1
cp gpurun_out/ffn_traffic.json profiles/ffn_traffic.json        # (bench.py reads the committed copy: on the box, this run's)
( time timeout 1200 python bench.py ) > gpurun_out/r06_bench_default.log 2>&1
echo "bench rc=$?"; grep '^{' gpurun_out/r06_bench_default.log | cut -c1-400
bash scripts/gpu_prof_graph.sh r06_graph > gpurun_out/r06_prof_graph.txt 2>&1; tail -3 gpurun_out/r06_prof_graph.txt | cut -c1-200
bash scripts/gpu_step_pmc.sh r06_step_pmc > gpurun_out/r06_step_pmc.txt 2>&1; head -5 gpurun_out/r06_step_pmc_summary.txt | cut -c1-200
timeout 400 bash scripts/ab.sh "DSVG_W_WARM=0 DSVG_GS_WARM=0" "DSVG_W_WARM=1" > gpurun_out/r06_ab_weight_warmup.log 2>&1
cat gpurun_out/r06_ab_weight_warmup.log
# ... and of the one-launch-per-stack group stages (second half of the round)
timeout 400 bash scripts/ab.sh "DSVG_GS_STACK=0" "DSVG_GS_STACK=1" > gpurun_out/r06_ab_gs_stack_final.log 2>&1
cat gpurun_out/r06_ab_gs_stack_final.log
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r06_pytest_gpu.log 2>&1; cat gpurun_out/r06_pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/r06_smoke.log 2>&1; cat gpurun_out/r06_smoke.log
