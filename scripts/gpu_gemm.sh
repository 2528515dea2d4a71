#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 300 python scripts/gemm_bench.py --dtype bf16 > gpurun_out/gemm_bench_bf16.log 2>&1
timeout 300 python scripts/gemm_bench.py --dtype fp32 --iters 5 > gpurun_out/gemm_bench_fp32.log 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o pmc1 -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "fwd" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o pmc2 -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "fwd" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/pmc1 gpurun_out/pmc2 | head -20 > gpurun_out/pmc_ls.log
