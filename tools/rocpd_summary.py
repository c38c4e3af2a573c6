#!/usr/bin/env python3
"""Summaries of rocprofv3's rocpd sqlite output (the default format of ROCm 7.2):
    rocpd_summary.py stats  <results.db> <out.csv>     per-kernel calls / total / average duration
    rocpd_summary.py pmc    <results.db> [name-substr] per-kernel mean counter values"""
import csv
import sqlite3
import sys


def main():
    mode, db = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    if mode == "stats":
        # one row per (kernel, grid, workgroup): the same template instance launched on different shapes
        # (block 1 at N = 4096 and the wide C = 128 / 256 points share tcn_wgrad_sw_kernel<2,5,2,6,false>) stays apart
        rows = list(c.execute(
            "select name, grid_x || 'x' || grid_y || 'x' || grid_z, workgroup_x, count(*), sum(duration) / 1e3, "
            "avg(duration) / 1e3, min(duration) / 1e3, max(duration) / 1e3 from kernels "
            "group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(duration) desc"))
        tot = sum(r[4] for r in rows) or 1.0
        rows = [tuple(r) + (round(100.0 * r[4] / tot, 2),) for r in rows]
        with open(sys.argv[3], "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Grid", "Workgroup", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
            w.writerows(rows)
        for r in rows[:30]:
            print("%-64s %-14s %5d %12.1f us avg %10.1f us %5.1f%%" % (
                r[0].replace("(anonymous namespace)::", "")[:64], r[1], r[3], r[4], r[5], r[8]))
    else:
        sub = sys.argv[3] if len(sys.argv) > 3 else ""
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "where kernel_name like ? group by kernel_name, counter_name")
        for r in c.execute(q, ("%" + sub + "%",)):
            print("%-70s %-12s n=%d mean=%.1f mean_dur_ns=%.0f" % (r[0][:70], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
