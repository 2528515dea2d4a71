#!/bin/bash
# HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) of the four launches the fused attention block replaces
# (scripts/attn_bench.py --pmc-unfused: 126976 rows dense S = 31, dropout 0.1, three repetitions)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
TAG=attnunf
cd /tmp
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG$1 -o p -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py --pmc-unfused > $GRAFT_REPO_ROOT/gpurun_out/$TAG$1.log 2>&1; }
run c "FETCH_SIZE"
run d "WRITE_SIZE"
cd $GRAFT_REPO_ROOT
python - > gpurun_out/${TAG}_summary.txt 2>&1 <<'PY'
import csv, collections, glob
tot = collections.defaultdict(float)
for f in sorted(glob.glob("gpurun_out/attnunf?/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if not any(t in k for t in ("ln_fwd", "gemm_bf16", "attn_fwd")): continue
        for c, x in v.items():
            print(k, c, f"avg per launch {sum(x)/len(x):.4e} KB, launches {len(x)}")
            tot[c] += sum(x) / 3.0          # three repetitions of the 4-launch sequence
print({c: f"{v:.4e} KB per attention sub-block forward" for c, v in tot.items()})
PY
cat gpurun_out/${TAG}_summary.txt | cut -c1-300
rm -rf gpurun_out/${TAG}c gpurun_out/${TAG}d
