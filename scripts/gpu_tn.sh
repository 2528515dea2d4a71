#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
echo "== 1D xcd-grouped" > gpurun_out/tn.log
timeout 200 python scripts/gemm_bench.py --dtype bf16 --only "dW1" >> gpurun_out/tn.log 2>&1
echo "== 2D" >> gpurun_out/tn.log
DSVG_SPLITK_2D=1 timeout 200 python scripts/gemm_bench.py --dtype bf16 --only "dW1" >> gpurun_out/tn.log 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o pmc3 -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "dW1   TN  512" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc4 -o pmc4 -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "dW1   TN  512" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc4.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc5 -o pmc5 -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "dW1   TN  512" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc5.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc6 -o pmc6 -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "dW1   TN  512" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc6.log 2>&1
