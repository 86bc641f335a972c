#!/bin/bash
# s = 9984 (sketchsize64 156), 12 000 genomes: kernel time and L2 fabric traffic with the block
# co-scheduling hint off / on (PPK_COSCHED).  FETCH_SIZE is reported in KB and, on gfx950, at half
# the bytes of a 16 B/lane stream (MI355X_MICROARCH.md, HBM): GB = 2 * KB / 1e6.
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in 0 1 2 3; do
  export PPK_COSCHED_DIAG=1
  if [ $c = 2 ]; then export PPK_COSCHED_SPIN0=400; fi
  if [ $c = 3 ]; then export PPK_COSCHED_SPIN0=2000 PPK_COSCHED_SPIN=100; fi
  OUT=gpurun_out/cosched$c; rm -rf $OUT; mkdir -p $OUT
  echo "== PPK_COSCHED=$c"
  PPK_COSCHED=$((c>0)) N=12000 ONLY5=1 timeout 300 python tools/ab_bigsketch.py 2>&1 | grep -v amdgpu.ids | tail -4
  PPK_COSCHED=$((c>0)) N=12000 ONLY5=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python tools/ab_bigsketch.py > /dev/null 2>&1
  PPK_COSCHED=$((c>0)) N=12000 ONLY5=1 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/f2 -o f -- python tools/ab_bigsketch.py > /dev/null 2>&1
  python3 - $OUT <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/f*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dist_kernel_v2" in r["Kernel_Name"] and "3, false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
f = agg.get("FETCH_SIZE", [0]); h = agg.get("TCC_HIT_sum", [0]); m = agg.get("TCC_MISS_sum", [1])
print("  FETCH_SIZE avg %.4g KB  => %.1f GB/launch from the fabric;  L2 hit rate %.3f  (n=%d launches)"
      % (sum(f) / len(f), 2 * sum(f) / len(f) / 1e6, sum(h) / (sum(h) + sum(m)), len(f)))
PY
done
