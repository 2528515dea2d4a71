#!/bin/bash
# PMC counters of EVERY kernel of the train step (bench.py, eager launches so that each launch is its own dispatch record; 2
# warm-up + 3 timed steps), in separate passes as the profiling guide prescribes (no trace domains besides --kernel-trace):
#   a) LDS bank conflicts / active LDS cycles, VALU and MFMA instruction counts, wave cycles
#   b) FETCH_SIZE   c) WRITE_SIZE   (KiB-units x 1024... summarised per kernel as MB per launch; FETCH_SIZE doubled for gfx950)
# usage: scripts/gpu_step_pmc.sh [tag]  ->  gpurun_out/<tag>_summary.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
TAG="${1:-steppmc}"
cd /tmp
run() { timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG$1 -o p -- python $GRAFT_REPO_ROOT/bench.py --graph 0 --steps 3 --warmup 2 --no-cpu-baseline --no-fp32 --no-torch-ref --no-roofline --no-extra-legs > $GRAFT_REPO_ROOT/gpurun_out/$TAG$1.log 2>&1; }
run a "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
run b "FETCH_SIZE"
run c "WRITE_SIZE"
cd $GRAFT_REPO_ROOT
python - "$TAG" > gpurun_out/${TAG}_summary.txt 2>&1 <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(f"gpurun_out/{tag}?/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
dur = collections.defaultdict(list)
for f in sorted(glob.glob(f"gpurun_out/{tag}a/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in agg.items():
    n = max(calls[k].values())
    g = lambda c: v.get(c, 0.0) / max(calls[k].get(c, 1), 1)
    # FETCH_SIZE / WRITE_SIZE: 1 KiB... the guide: units of 64 B?  keep the raw sums and the guide's conversion:
    fetch_mb = g("FETCH_SIZE") * 1024 * 2 / 1e6       # KiB units, doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
    write_mb = g("WRITE_SIZE") * 1024 / 1e6
    t = dur.get(k, [])
    avg = sum(t) / len(t) if t else 0.0
    rows.append((sum(t), k[:90], n, avg, fetch_mb, write_mb, g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE"),
                 g("SQ_INSTS_VALU"), g("SQ_INSTS_MFMA"), g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES")))
rows.sort(reverse=True)
# SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over every SIMD of the chip (32 per 32x32x16 bf16 MFMA,
# MI355X_MICROARCH.md); the fraction of the chip's matrix-pipe time = that / (1024 SIMDs x launch duration x clock).  The clock is
# taken as 2.4 GHz, so under power management (1.5-2.3 GHz inside the fused kernels) the column UNDERSTATES the busy fraction at
# the clock the launch ran at by up to the same ratio; round 4's column divided by 4 x SQ_BUSY_CYCLES (one count per SE, not
# per SIMD) and came out above 1.
def mfma_busy(cycles, avg_us):
    return cycles / (1024 * avg_us * 2400.0) if avg_us else 0.0
print("kernel,launches,avg_us,fetch_MB_per_launch,write_MB_per_launch,GBps,lds_conflict_frac,valu_per_mfma,mfma_busy_frac_at_2p4GHz")
for tot, k, n, avg, fm, wm, bc, ia, nv, nm, mb, busy in rows[:60]:
    gbps = (fm + wm) / avg * 1e3 if avg else 0
    print(f"{k},{n},{avg:.1f},{fm:.1f},{wm:.1f},{gbps:.0f},{(bc / ia if ia else 0):.3f},{(nv / nm if nm else 0):.1f},{mfma_busy(mb, avg):.3f}")
PY
cut -c1-220 gpurun_out/${TAG}_summary.txt | head -64
rm -rf gpurun_out/${TAG}a gpurun_out/${TAG}b gpurun_out/${TAG}c
