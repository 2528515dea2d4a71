#!/bin/bash
# Round-4 review item 3: what clock do the kernels of the train step run at?
#   1. scripts/clock_probe.py: sclk samples + ms/step per 25-step window over 10 s (ramp or stationary?)
#   2. rocprofv3 --pmc GRBM_GUI_ACTIVE (+ SQ busy / MFMA busy) with --kernel-trace over the eager step: effective clock per
#      kernel = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS)
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_clock.sh'  -> gpurun_out/clock_*.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 300 python scripts/clock_probe.py > gpurun_out/clock_probe.log 2>&1
echo "clock probe rc=$?"; cat gpurun_out/clock_probe.log | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv \
    -d $GRAFT_REPO_ROOT/gpurun_out/clockpmc -o p -- python $GRAFT_REPO_ROOT/bench.py --graph 0 --steps 6 --warmup 3 \
    --no-cpu-baseline --no-fp32 --no-torch-ref --no-roofline --no-extra-legs > $GRAFT_REPO_ROOT/gpurun_out/clock_pmc.log 2>&1
echo "pmc rc=$?"; tail -n 3 $GRAFT_REPO_ROOT/gpurun_out/clock_pmc.log | cut -c1-300
cd $GRAFT_REPO_ROOT
python - > gpurun_out/clock_pmc_summary.csv 2>&1 <<'PY'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
disp = {}
for f in sorted(glob.glob("gpurun_out/clockpmc/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp[r["Dispatch_Id"]]["_k"] = r["Kernel_Name"]
dur = {}
for f in sorted(glob.glob("gpurun_out/clockpmc/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
per = collections.defaultdict(list)
for d, v in disp.items():
    if d in dur and dur[d] > 0:
        per[v["_k"][:80]].append((dur[d], v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CYCLES", 0.0)))
rows = []
for k, v in per.items():
    t = sum(x[0] for x in v)
    g = sum(x[1] for x in v)
    m = sum(x[2] for x in v)
    rows.append((t, k, len(v), t / len(v), g / t / 8 / 1e3, m / 1024.0 / max(g / 8, 1.0)))
rows.sort(reverse=True)
print("kernel,launches,avg_us,effective_clock_GHz(GRBM_GUI_ACTIVE/8/duration),mfma_busy_frac_of_clock_cycles")
for t, k, n, avg, ghz, mb in rows[:40]:
    print(f"\"{k}\",{n},{avg:.1f},{ghz:.3f},{mb:.3f}")
PY
head -30 gpurun_out/clock_pmc_summary.csv | cut -c1-200
rm -rf gpurun_out/clockpmc
