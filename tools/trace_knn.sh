#!/bin/bash
# Kernel timeline of ONE "neighbours from tiles" call (10 neighbours of each of 10 000 genomes, engine.knn_from_sketches(method="tiles")):
# every launch of the last call with its duration and the gap to the launch before it.
OUT=gpurun_out/knntrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
cat > $OUT/run.py <<'PY'
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from poppunk_amd import engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(10000, K)
db = engine.SketchDB(sk, 16, 14)
for _ in range(6):
    t0 = time.perf_counter(); r = engine.knn_from_sketches(db, K, T, 10, method="tiles"); torch.cuda.synchronize(); t1 = time.perf_counter()
print("wall %.3f ms" % ((t1 - t0) * 1e3))
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o b -- python $OUT/run.py > $OUT/run.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/knntrace/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "dist_kernel_v2" in r["Kernel_Name"]]
s = starts[-1]
# walk back to the first launch of this call (a state-init kernel precedes the distance kernel)
while s > 0 and int(rows[s]["Start_Timestamp"]) - int(rows[s - 1]["End_Timestamp"]) < 60000 and "dist_kernel_v2" not in rows[s - 1]["Kernel_Name"]: s -= 1
t0 = int(rows[s]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[s:]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  +%8.1f us  gap %6.1f  run %7.1f  %s" % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, r["Kernel_Name"][:100]))
    prev = b; busy += b - a
print("  span %.1f us, kernels %.1f us" % ((prev - t0) / 1e3, busy / 1e3))
PY
tail -3 $OUT/run.log
