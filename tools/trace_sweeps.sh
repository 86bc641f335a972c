#!/bin/bash
# Kernel timeline of ONE thresholdIterate1D / 2D call on the 10 000-genome matrix: every launch of the call with its
# duration and the gap to the launch before it (rocprofv3 --kernel-trace over tools/time_sweeps.py; the last call of each).
OUT=gpurun_out/sweeptrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o b -- python tools/time_sweeps.py > $OUT/run.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/sweeptrace/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def call(first_kernel, last_kernel):
    ends = [i for i, r in enumerate(rows) if last_kernel in r["Kernel_Name"]]
    e = ends[-1]
    s = max(i for i in range(e) if first_kernel in rows[i]["Kernel_Name"])
    t0 = int(rows[s]["Start_Timestamp"]); prev = t0; busy = 0
    for r in rows[s:e + 1]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("  +%7.1f us  gap %5.1f  run %6.1f  grid %-9s %s" % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"][:90]))
        prev = b; busy += b - a
    print("  span %.1f us, kernels %.1f us" % ((prev - t0) / 1e3, busy / 1e3))
print("1-D"); call("ti1_classify", "ti1_emit")
print("2-D"); call("ti1_classify", "ti2_emit")
PY
tail -5 $OUT/run.log
