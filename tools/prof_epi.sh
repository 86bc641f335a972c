#!/bin/bash
# What the epilogue of dist_kernel_v2 executes: SQ instruction counters of the 10 240^2 job with and
# without it (ablate 0 / 1), one --pmc pass each.
set -u
OUT=gpurun_out/epi
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY"
for a in ${ABLATES:-0 1}; do
  PPK_ABLATE=$a REPS=3 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$a -o r -- python tools/ab_square.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
res = {}
import os
AB = [int(x) for x in os.environ.get("ABLATES", "0 1").split()]
for a in AB:
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/epi/pmc_%d/*counter_collection.csv" % a):
        for r in csv.DictReader(open(f)):
            if "dist_kernel_v2" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[a] = {k: sum(v) / len(v) for k, v in agg.items()}
res = {0: res[AB[0]], 1: res[AB[1]]}
pairs = 10240.0 * 10240.0
print("%-18s %14s %14s %12s" % ("counter", "with epilogue", "without", "epilogue/pair (x64 lanes)"))
for k in sorted(res[0]):
    d = res[0][k] - res[1].get(k, 0.0)
    print("%-18s %14.0f %14.0f %12.2f" % (k, res[0][k], res[1].get(k, 0.0), d * 64 / pairs))
PY
