#!/usr/bin/env python
"""Sum the FETCH_SIZE / WRITE_SIZE counters (KB) of the GEMM dispatches issued by `bench.py --ffn-replay N` (they are the
LAST launches of the run: N x launches_per_step GEMM dispatches, split-K partial reductions in between) -> JSON."""
import csv
import glob
import json
import re
import sys


def last_gemm_dispatches(d, counter, want):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    gemm = [r for r in rows if "gemm" in r["Kernel_Name"]]
    sel = gemm[-want:]
    return sum(float(r["Counter_Value"]) for r in sel), len(sel)


def main():
    n, dfetch, dwrite, log = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
    txt = open(log).read()
    m = re.search(r"'launches_per_step': (\d+).*?'executed_gflop_per_step': ([0-9.]+)", txt)
    per_step, gflop = int(m.group(1)), float(m.group(2))
    dtype = re.search(r'"dtype": "(\w+)"', txt).group(1)
    fetch_kb, nf = last_gemm_dispatches(dfetch, "FETCH_SIZE", n * per_step)
    write_kb, nw = last_gemm_dispatches(dwrite, "WRITE_SIZE", n * per_step)
    assert nf == nw == n * per_step, (nf, nw, n * per_step)
    # gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes for 16-byte-per-lane streaming reads -> doubled
    # (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is; both in KB
    # kernel-trace durations of the same dispatches (agreement check for bench.py's HIP-event time)
    tr = glob.glob(dfetch + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(tr)) if "gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-n * per_step:]]
    fetch_gb = 2.0 * fetch_kb * 1024 / n / 1e9
    write_gb = write_kb * 1024 / n / 1e9
    print(json.dumps({"executed_gflop_per_step": gflop, "dtype": dtype, "launches_per_step": per_step,
                      "fetch_GB_per_step": round(fetch_gb, 3), "write_GB_per_step": round(write_gb, 3),
                      "hbm_GB_per_step": round(fetch_gb + write_gb, 3),
                      "rocprofv3_kernel_trace_avg_gemm_launch_us": round(sum(dur) / len(dur), 2),
                      "rocprofv3_kernel_trace_ffn_gemm_ms_per_step": round(sum(dur) / n / 1e3, 3),
                      "source": "scripts/gpu_ffn_traffic.sh, FETCH_SIZE x2 + WRITE_SIZE over %d replays" % n}))


if __name__ == "__main__":
    main()
