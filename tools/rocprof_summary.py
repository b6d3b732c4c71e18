#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (*_results.db, the default output format of
ROCm 7.2's `rocprofv3 --kernel-trace --stats`) into the per-kernel summary text
we commit under profiles/.   usage: rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("""select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3,
                         max(end-start)/1e3, max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x)
                         from kernels group by name order by 3 desc""").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    out = ["%-72s %6s %12s %10s %10s %10s %6s %5s %5s %6s %10s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "grid_x", "wg")]
    for r in rows:
        out.append("%-72s %6d %12.1f %10.1f %10.1f %10.1f %6.2f %5d %5d %6d %10d %5d" % (
            r[0][:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
