#!/bin/bash
# per-launch durations of a long run (DVFS ramp / throttle behaviour)
OUT=gpurun_out/trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o b -- python bench.py --no-cpu --warmup 5 --steps 300 > $OUT/bench.log 2>&1
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/trace/kt/b_kernel_trace.csv")) if "dist_kernel_v2" in r["Kernel_Name"]]
t0=int(rows[0]["Start_Timestamp"])
for i,r in enumerate(rows):
    if i<40 or i%10==0:
        print("%4d start %8.1f ms dur %.3f ms" % (i,(int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
