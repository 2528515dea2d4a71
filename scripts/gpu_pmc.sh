#!/bin/bash
# PMC counters for selected gemm_bench cases: usage gpu_pmc.sh "<only-filter>" [tag] [impls]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
ONLY="$1"; TAG="${2:-pmc}"; IMPLS="${3:-0}"
cd /tmp
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG$1 -o p -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py --dtype bf16 --only "$ONLY" --impls "$IMPLS" --iters 2 > $GRAFT_REPO_ROOT/gpurun_out/$TAG$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM"
run c "FETCH_SIZE WRITE_SIZE"
run d "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $TAG > gpurun_out/${TAG}_summary.txt 2>&1
cat gpurun_out/${TAG}_summary.txt | cut -c1-700
rm -rf gpurun_out/${TAG}a gpurun_out/${TAG}b gpurun_out/${TAG}c gpurun_out/${TAG}d
