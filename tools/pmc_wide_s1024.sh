# PMC counters of the windowed ("wide") tile kernel at s = 1 024, 10 000 genomes, 17 and 21 k-mer lengths
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_wide; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from poppunk_amd import _lib, engine, synth
nk = int(sys.argv[1]); n = 10000
kmers = np.arange(31 - nk + 1, 32, dtype=np.int32)
t = synth.make_sketches_device(n, kmers, sketchsize64=16, seed=7, device="cuda:0", chunk=8192)
db = engine.SketchDB(t, 16, 14); del t
tbl = synth.random_match_table(kmers)
out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.float32, device="cuda")
_lib.set_option("ksplit", 0)
for _ in range(3):
    engine.dist(db, None, kmers, tbl, out=out)
torch.cuda.synchronize()
PY
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for nk in 17 21; do
    rocprofv3 --pmc $grp --output-format csv -d $OUT/p${i}_$nk -o p -- python $OUT/run.py $nk > $OUT/p${i}_$nk.txt 2>&1 || echo "pass $i failed: $grp"
  done
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_wide/**/*counter_collection.csv", recursive=True):
    nk = f.split("/p")[1].split("_")[1].split("/")[0]
    for r in csv.DictReader(open(f)):
        if "dist_kernel_v2" in r["Kernel_Name"]:
            agg[nk][r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(set(agg["17"]) | set(agg["21"])):
    a = agg["17"].get(c, [0]); b = agg["21"].get(c, [0])
    print("%-22s nk=17 %.5g   nk=21 %.5g   ratio %.3f (21/17 = 1.235)" % (c, sum(a)/len(a), sum(b)/len(b), (sum(b)/len(b)) / max(sum(a)/len(a), 1e-9)))
PY
