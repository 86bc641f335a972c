#!/bin/bash
# What sits between two consecutive launches of the dominant kernel in bench.py's timed loop:
# kernel-trace of 100 steps, gap = next start - this end, and the other kernels launched in between.
OUT=gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o b -- python bench.py --no-cpu --no-host-call --no-config5 --warmup 20 --steps 100 > $OUT/bench.log 2>&1
python - <<'PY'
import csv, collections
rows = sorted(csv.DictReader(open("gpurun_out/gaps/kt/b_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "dist_kernel_v2" in r["Kernel_Name"]]
idx = idx[-100:]
gaps, durs, between = [], [], collections.Counter()
for a, b in zip(idx[:-1], idx[1:]):
    gaps.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
    durs.append((int(rows[a]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
    for r in rows[a + 1:b]:
        between[r["Kernel_Name"][:60]] += 1
gaps.sort()
print("launches %d  kernel avg %.1f us  gap median %.1f us  mean %.1f us  p90 %.1f us  (= %.2f %% of a step)"
      % (len(idx), sum(durs) / len(durs), gaps[len(gaps) // 2], sum(gaps) / len(gaps), gaps[int(len(gaps) * .9)],
         100 * sum(gaps) / (sum(gaps) + sum(durs))))
for k, v in between.most_common(5):
    print("  between two launches: %s x %.2f per step" % (k, v / len(gaps)))
PY
tail -c 400 $OUT/bench.log
