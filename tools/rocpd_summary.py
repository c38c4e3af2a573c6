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
        rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                              "order by total_duration desc"))
        with open(sys.argv[3], "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            w.writerows(rows)
        for r in rows[:25]:
            print("%-80s %5d %12.1f us avg %10.1f us %5.1f%%" % (r[0].replace("(anonymous namespace)::", "")[:80],
                                                                 r[1], r[2], r[3], r[4]))
    else:
        sub = sys.argv[3] if len(sys.argv) > 3 else ""
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "where kernel_name like ? group by kernel_name, counter_name")
        for r in c.execute(q, ("%" + sub + "%",)):
            print("%-70s %-12s n=%d mean=%.1f mean_dur_ns=%.0f" % (r[0][:70], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
