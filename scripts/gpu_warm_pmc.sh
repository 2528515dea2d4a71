#!/bin/bash
# Round 6 evidence for the weight warm-up: L2 hit / miss counts (TCC_HIT_sum, TCC_MISS_sum) and durations of the kernels that stream a
# weight image, with the up-front request off (DSVG_W_WARM=0 DSVG_GS_WARM=0) and on - eager launches of the train step, 2 warm-up + 3
# timed steps per setting (counters in their own rocprofv3 pass with --kernel-trace only).
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_warm_pmc.sh'  -> gpurun_out/r06_warm_pmc_summary.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
cd /tmp
for w in 0 1; do
  DSVG_W_WARM=$w DSVG_GS_WARM=$w timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/warmpmc$w -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --graph 0 --steps 3 --warmup 2 --no-cpu-baseline --no-fp32 --no-torch-ref --no-roofline --no-extra-legs > $GRAFT_REPO_ROOT/gpurun_out/warmpmc$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - > gpurun_out/r06_warm_pmc_summary.txt 2>&1 <<'PY'
import csv, collections, glob
names = ("gs_layer_fwd_kernel", "gs_layer_bwd_kernel", "ffn_fwd_kernel", "attn_block_fwd_kernel", "ffn_bwd_dx_kernel")
print("kernel,warm,launches,avg_us,L2_hit_M_per_launch,L2_miss_M_per_launch,hit_rate")
for w in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for f in sorted(glob.glob(f"gpurun_out/warmpmc{w}/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = next((n for n in names if n in r["Kernel_Name"]), None)
            if k is None:
                continue
            if "attn_block_fwd" in k:
                k += "<tiled>" if "true, true" in r["Kernel_Name"] else "<dense>"
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "TCC_HIT_sum":
                cnt[k] += 1
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(f"gpurun_out/warmpmc{w}/**/*kernel_trace.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = next((n for n in names if n in r["Kernel_Name"]), None)
            if k is None:
                continue
            if "attn_block_fwd" in k:
                k += "<tiled>" if "true, true" in r["Kernel_Name"] else "<dense>"
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(agg):
        n = max(cnt[k], 1)
        h, m = agg[k]["TCC_HIT_sum"] / n / 1e6, agg[k]["TCC_MISS_sum"] / n / 1e6
        d = sum(dur[k]) / max(len(dur[k]), 1)
        print(f"{k},{w},{n},{d:.1f},{h:.2f},{m:.2f},{h / (h + m) if h + m else 0:.3f}")
PY
cat gpurun_out/r06_warm_pmc_summary.txt
rm -rf gpurun_out/warmpmc0 gpurun_out/warmpmc1
