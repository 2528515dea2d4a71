#!/bin/bash
# matrix-pipe busy cycles, wave cycles, LDS conflicts of ffn_fwd_rs_kernel next to ffn_fwd_kernel<4, ., true> (round 5):
# rocprofv3 --pmc in separate passes with --kernel-trace only -> gpurun_out/r05_ffn_rs_pmc_summary.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
TAG=r05_ffn_rs_pmc
cd /tmp
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG$1 -o p -- python $GRAFT_REPO_ROOT/scripts/ffn_rs_pmc_run.py > $GRAFT_REPO_ROOT/gpurun_out/$TAG$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"
cd $GRAFT_REPO_ROOT
python - > gpurun_out/${TAG}_summary.txt 2>&1 <<'PY'
import csv, collections, glob
disp = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/r05_ffn_rs_pmc?/**/*counter_collection.csv", recursive=True)):
    tag = f.split("r05_ffn_rs_pmc")[1][0]
    for r in csv.DictReader(open(f)):
        if "ffn_fwd" not in r["Kernel_Name"]: continue
        disp[(tag, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[(tag, r["Dispatch_Id"])]["_k"] = r["Kernel_Name"]
        disp[(tag, r["Dispatch_Id"])]["_grid"] = int(r.get("Grid_Size", 0) or 0)
dur = {}
for f in sorted(glob.glob("gpurun_out/r05_ffn_rs_pmc?/**/*kernel_trace.csv", recursive=True)):
    tag = f.split("r05_ffn_rs_pmc")[1][0]
    for r in csv.DictReader(open(f)):
        dur[(tag, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for key, v in disp.items():
    k = v["_k"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    name = f"{k} grid={v['_grid']}"
    for c, x in v.items():
        if not c.startswith("_"): agg[name][c].append(x)
    if key in dur: agg[name]["dur_us_" + key[0]].append(dur[key])
print("kernel,launches,avg_us,mfma_busy_frac_of_2.4GHz_cycles,mfma_busy_frac_of_own_clock,eff_clock_GHz,wave_cycles_per_launch(quad),lds_conflict_frac,valu_per_mfma")
for name, v in sorted(agg.items()):
    last = lambda c: (sum(v[c][-3:]) / max(len(v[c][-3:]), 1)) if v.get(c) else 0.0
    us = last("dur_us_a")
    busy, gui = last("SQ_VALU_MFMA_BUSY_CYCLES"), last("GRBM_GUI_ACTIVE")
    clk = gui / 8 / us / 1e3 if us else 0
    print(f"{name},{len(v.get('dur_us_a', []))},{us:.1f},{busy / (1024 * us * 2400) if us else 0:.3f},{busy / 1024 / (gui / 8) if gui else 0:.3f},{clk:.2f},"
          f"{last('SQ_WAVE_CYCLES'):.3e},{(last('SQ_LDS_BANK_CONFLICT') / last('SQ_LDS_IDX_ACTIVE')) if last('SQ_LDS_IDX_ACTIVE') else 0:.3f},"
          f"{(last('SQ_INSTS_VALU') / last('SQ_INSTS_MFMA')) if last('SQ_INSTS_MFMA') else 0:.1f}")
PY
cat gpurun_out/${TAG}_summary.txt | cut -c1-250
rm -rf gpurun_out/${TAG}a gpurun_out/${TAG}b
