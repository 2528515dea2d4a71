#!/bin/bash
# HBM traffic of ffn_fwd_kernel (the kernel bench.py's roofline.traffic is quoted on): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
# separate passes (with --kernel-trace only) over scripts/ffn_traffic_run.py, summarised into profiles/ffn_traffic.json with the
# commit and the hash of the kernel source it was taken on.  FETCH_SIZE is calibrated on the inference variant of the same
# kernel (known read bytes) instead of the blanket x2 of MI355X_MICROARCH.md, which holds for wide coalesced streaming reads.
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_ffn_traffic.sh'   -> gpurun_out/ffn_traffic.json (copy to profiles/)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ffntr_$C -o p -- \
      python $GRAFT_REPO_ROOT/scripts/ffn_traffic_run.py > $GRAFT_REPO_ROOT/gpurun_out/ffntr_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - > gpurun_out/ffn_traffic.json <<'PY'
import csv, glob, json, subprocess, sys
sys.path.insert(0, ".")
from bench import kernel_source_hash
def per_dispatch(counter):
    f = glob.glob(f"gpurun_out/ffntr_{counter}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "ffn_fwd_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # template arguments <ring slots, TRAIN, packed activation code>: the second one tells the variant
    return [(r["Kernel_Name"].split("<")[1].split(">")[0].split(",")[1].strip() == "true", float(r["Counter_Value"])) for r in rows]
def durations():
    f = glob.glob("gpurun_out/ffntr_FETCH_SIZE/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "ffn_fwd_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
fe, wr, du = per_dispatch("FETCH_SIZE"), per_dispatch("WRITE_SIZE"), durations()
assert len(fe) == len(wr) == len(du) == 20, (len(fe), len(wr), len(du))
out = {"dtype": "bf16"}
for k, rows in enumerate((63488, 40960)):
    inf = slice(10 * k + 2, 10 * k + 5)          # last 3 of the 5 inference launches
    tr = slice(10 * k + 7, 10 * k + 10)
    assert not any(t for t, _ in fe[inf]) and all(t for t, _ in fe[tr])
    mean = lambda v: sum(x for _, x in v) / len(v)
    f_inf, f_tr = mean(fe[inf]) * 1024 / 1e6, mean(fe[tr]) * 1024 / 1e6          # KiB units -> MB, raw
    w_inf, w_tr = mean(wr[inf]) * 1024 / 1e6, mean(wr[tr]) * 1024 / 1e6
    known_read = (rows * 512 + 512 * 1024) / 1e6
    cal = known_read / f_inf
    out[f"rows_{rows}"] = {
        "inference": {"fetch_raw_MB": round(f_inf, 1), "write_MB": round(w_inf, 1), "known_read_MB": round(known_read, 1),
                      "known_write_MB": round(rows * 512 / 1e6, 1), "avg_us": round(sum(du[inf]) / 3, 1)},
        "fetch_calibration": round(cal, 3),
        "training": {"fetch_raw_MB": round(f_tr, 1), "fetch_MB": round(f_tr * cal, 1), "write_MB": round(w_tr, 1),
                     "MB_per_launch": round(f_tr * cal + w_tr, 1), "avg_us": round(sum(du[tr]) / 3, 1),
                     "algorithmic_MB_training_variant": round(rows * (512 + 512 + 1024 + 512 + 4) / 1e6, 1),
                     "fused_algorithmic_MB": round(rows * 1024 / 1e6, 1)}}
big = out["rows_63488"]["training"]
out.update({
    "kernel": "ffn_fwd_kernel<4, true, true> (the default: 256-row workgroups, packed activation code; training variant: y + h + xh + rstd out), 63,488 rows = the largest launch of the step, "
              "the rows evicted from every cache before each launch",
    "MB_per_launch": big["MB_per_launch"], "fetch_MB": big["fetch_MB"], "write_MB": big["write_MB"],
    "avg_launch_us": big["avg_us"], "fused_algorithmic_MB": big["fused_algorithmic_MB"],
    "over_fused_algorithmic": round(big["MB_per_launch"] / big["fused_algorithmic_MB"], 2),
    "expected_MB_training_variant": big["algorithmic_MB_training_variant"],
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None,
    "ffn_fused_hip_code_sha256_16": kernel_source_hash("deepsvg_amd/csrc/ffn_fused.hip"),
    "source": "scripts/gpu_ffn_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over scripts/ffn_traffic_run.py; "
              "FETCH_SIZE calibrated on the inference variant of the same kernel: known read bytes / reported)"})
print(json.dumps(out, indent=1))
PY
cat gpurun_out/ffn_traffic.json | head -60
rm -rf gpurun_out/ffntr_FETCH_SIZE gpurun_out/ffntr_WRITE_SIZE
