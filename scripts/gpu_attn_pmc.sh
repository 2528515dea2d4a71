#!/bin/bash
# PMC counters of the fused attention block (scripts/attn_bench.py --pmc: 126976 rows dense S = 31, dropout 0.1, three launches per variant)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
TAG="${1:-attnpmc}"
cd /tmp
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG$1 -o p -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py --pmc > $GRAFT_REPO_ROOT/gpurun_out/$TAG$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
cd $GRAFT_REPO_ROOT
python - "$TAG" > gpurun_out/${TAG}_summary.txt 2>&1 <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/{tag}?/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "attn_block" not in k: continue
        print(k, {c: f"{sum(x)/len(x):.4e}" for c, x in v.items()}, "launches", len(next(iter(v.values()))))
for f in sorted(glob.glob(f"gpurun_out/{tag}a/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "attn_block" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"][:75]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in by.items():
        print("trace", k, f"n={len(v)} avg {sum(v)/len(v):.1f} us min {min(v):.1f}", "vgpr", rows[0].get("VGPR_Count"), "lds", rows[0].get("LDS_Block_Size"))
PY
cat gpurun_out/${TAG}_summary.txt | cut -c1-900
rm -rf gpurun_out/${TAG}a gpurun_out/${TAG}b gpurun_out/${TAG}c gpurun_out/${TAG}d
