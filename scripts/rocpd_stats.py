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
    txt = "\n".join(lines) + "\n"
    if out_csv:
        open(out_csv, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
