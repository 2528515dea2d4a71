#!/usr/bin/env python
"""Per-kernel summary (calls, total, avg/min/max duration, share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db in this ROCm build)."""
import re
import sqlite3
import sys


def main(db_path, out_csv=None, skip_calls=0):
    db = sqlite3.connect(db_path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                     f"max(d.end-d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size) "
                     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes"]
    for r in rows:
        name = re.sub(r"\(.*", "", r[0]).replace(",", ";")[:90]
        lines.append(f"{name},{r[1]},{r[2] / 1e6:.3f},{r[3] / 1e3:.1f},{r[4] / 1e3:.1f},{r[5] / 1e3:.1f},"
                     f"{100 * r[2] / tot:.1f},{r[6]},{r[7]},{r[8]}")
    lines.append(f"TOTAL,,{tot / 1e6:.3f},,,,100.0,,,")
    # where the time goes by launch duration (launch-latency-bound tail vs bandwidth-bound body)
    edges = [0, 5, 10, 20, 50, 100, 200, 1e9]
    durs = [r[0] / 1e3 for r in c.execute(f"select end-start from {kd}")]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = [d for d in durs if lo <= d < hi]
        lines.append(f"HIST {lo:g}-{hi:g} us,{len(sel)},{sum(sel) / 1e3:.3f},,,,{100 * sum(sel) * 1e3 / tot:.1f},,,")
    # timeline of the LAST step (between the last two adamw launches): launch order, duration and the idle gap before it
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    tl = c.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    marks = [i for i, r in enumerate(tl) if "adamw" in r[0]]
    if len(marks) >= 2 and out_csv:
        seg = tl[marks[-2] + 1:marks[-1] + 1]
        with open(out_csv.replace(".csv", "_timeline.csv"), "w") as f:
            f.write("idx,kernel,start_us,dur_us,gap_before_us,queue\n")
            prev_end = seg[0][1]
            for i, (nm, st, en, qu) in enumerate(seg):
                nm = re.sub(r"\(.*", "", nm).replace(",", ";")
                nm = re.sub(r"_ZN12_GLOBAL__N_1\d+|void |\(anonymous namespace\)::", "", nm)[:60]
                f.write(f"{i},{nm},{(st - seg[0][1]) / 1e3:.1f},{(en - st) / 1e3:.1f},{(st - prev_end) / 1e3:.1f},{qu}\n")
                prev_end = max(prev_end, en)
    txt = "\n".join(lines) + "\n"
    if out_csv:
        open(out_csv, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
