#!/usr/bin/env python3
"""Summarise rocprofv3 sqlite outputs: per-kernel durations and PMC counter averages.

usage: python tools/pmc_summary.py <dir-with-*_results.db> [...]
"""
import glob
import os
import sqlite3
import sys

for path in sys.argv[1:]:
    for db in sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        print("==", db)
        try:
            for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
                print("  kernel %-64s calls=%-4d avg=%.1f us  %.1f%%" % (r[0][:64], r[1], r[3] / 1e3, r[4]))
        except Exception as e:
            print("  (no top_kernels)", e)
        try:
            q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                 "where kernel_name like '%dist_kernel%' or kernel_name like '%mask_%' "
                 "or kernel_name like '%assign_kernel%' group by kernel_name, counter_name")
            for r in cur.execute(q):
                print("  %-36s %-26s %.5g (n=%d)" % (r[0][5:41], r[1], r[2], r[3]))
        except Exception:
            pass
