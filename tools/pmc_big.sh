set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/big; rm -rf $OUT; mkdir -p $OUT
N=12000 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python tools/ab_bigsketch.py > /dev/null 2>&1
N=12000 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/w -o w -- python tools/ab_bigsketch.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("f","w"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/big/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "dist_kernel_v2" in r["Kernel_Name"] and "3, false" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, "n=%d" % len(v), "avg %.4g" % (sum(v)/len(v)), "max %.4g" % max(v))
PY
