#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) as text: per-kernel launch statistics
(the `--stats` view) and, when present, PMC counter sums per kernel.  Usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        try:
            rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                               "max(grid_x), max(workgroup_x), max(lds_size), max(scratch_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) "
                               "from kernels group by name order by sum(end-start) desc").fetchall()
        except sqlite3.OperationalError:
            cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
            print("kernels view columns:", cols)
            raise
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}  grid wg lds scratch vgpr agpr sgpr")
        for r in rows:
            name = r[0][:60]
            print(f"{name:60s} {r[1]:6d} {r[2]/1e6:10.3f} {r[3]/1e3:10.2f} {r[4]/1e3:10.2f} {r[5]/1e3:10.2f} {100*r[2]/tot:6.2f}  "
                  f"{r[6]} {r[7]} {r[8]} {r[9]} {r[10]} {r[11]} {r[12]}")
        try:
            pmc = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                              "group by kernel_name, counter_name order by kernel_name").fetchall()
            if pmc:
                print(f"{'kernel':60s} {'counter':>14s} {'dispatches':>10s} {'sum':>16s} {'avg/dispatch':>16s}")
                for r in pmc:
                    print(f"{r[0][:60]:60s} {r[1]:>14s} {r[2]:10d} {r[3]:16.1f} {r[4]:16.1f}")
        except sqlite3.OperationalError as e:
            print("no counters:", e)


if __name__ == "__main__":
    main()
