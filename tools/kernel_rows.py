#!/usr/bin/env python3
"""Per-(kernel, grid) rows of a rocprofv3 --kernel-trace CSV: calls, average / min / max duration.

rocprofv3's own --stats summary has one row per kernel NAME; bench.py launches the same instantiation on
differently sized jobs (10 000 self and 50 000 x 10 000 both run dist_kernel_v2<8,0,2,false>), so the rows that
bench.py's `kernel_ms` figures are checked against are split by grid here.

    python tools/kernel_rows.py <..._kernel_trace.csv>  >  profiles/r04/bench_kernel_rows.csv
"""
import collections
import csv
import sys

rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    full = r["Kernel_Name"]
    anon = "(anonymous namespace)::"
    name = (anon + full[len(anon):].split("(")[0]) if full.startswith(anon) else full.split("(")[0]
    if not name:
        name = full[:80]
    grid = "x".join(r.get(k, "1") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = "x".join(r.get(k, "1") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows.setdefault((name, grid, wg), []).append(d)
w = csv.writer(sys.stdout)
w.writerow(["Name", "Grid(work-items)", "Workgroup", "Calls", "AverageNs", "MinNs", "MaxNs", "TotalNs"])
for (name, grid, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, wg, len(v), "%.1f" % (sum(v) / len(v)), min(v), max(v), sum(v)])
