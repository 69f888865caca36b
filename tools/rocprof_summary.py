#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) per kernel.

usage: tools/rocprof_summary.py <results.db> [title]  > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    rows = db.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3,"
        " max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x)"
        " from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {title}")
    print(f"# rocprofv3 --kernel-trace --stats; durations in microseconds; {len(rows)} distinct kernels, "
          f"total GPU kernel time {tot/1e3:.2f} ms")
    print(f"{'kernel':78s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} "
          f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'grid':>9s} {'wg':>4s}")
    for r in rows[:30]:
        print(f"{r[0][:78]:78s} {r[1]:6d} {r[2]:11.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {100*r[2]/tot:6.2f} "
              f"{r[6]:5d} {r[7]:5d} {r[8]:6d} {r[9]:9d} {r[10]:4d}")


if __name__ == "__main__":
    main()
