#!/bin/bash
OUT=gpurun_out/epi; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/pmc -o u -- ./tools/ubench_pipe.out > /dev/null 2>&1
python - <<'PY'
import csv, collections
rows=[r for r in csv.DictReader(open("gpurun_out/epi/pmc/u_counter_collection.csv")) if "dist_kernel_v2" in r["Kernel_Name"]]
by=collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
ids=list(by)
print(len(ids),"dispatches")
for i,d in enumerate(ids):
    if i%6==1: print(i//6, {k:round(v/1e6,2) for k,v in by[d].items()})
PY
