#!/bin/bash
# PMC counters for selected gemm_bench cases: usage gpu_pmc.sh "<only-filter>" [tag]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
ONLY="$1"; TAG="${2:-pmc}"
cd /tmp
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG$1 -o p -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "$ONLY" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/$TAG$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SALU"
run c "FETCH_SIZE"
run d "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
