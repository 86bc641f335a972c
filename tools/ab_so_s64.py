"""Same-box A/B of builds of libppk_hip.so at two sketch sizes: kernel-1 time of N genomes self at s = 1 024 and at
the default s = 9 984, builds alternating in separate processes.

    python tools/ab_so_s64.py A.so B.so [...] [rounds]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from poppunk_amd import _lib
_lib.SO_PATH = os.path.abspath(sys.argv[1])
import torch
from poppunk_amd import engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
n = int(os.environ.get("N", "10000"))
res = []
for s64, warm, reps in ((16, 40, 60), (156, 4, 6)):
    db = engine.SketchDB(synth.make_sketches_device(n, K, sketchsize64=s64, device="cuda:0"), s64, 14, device=0)
    buf = None
    for _ in range(warm):
        buf, _f = engine.dist(db, None, K, T, out=buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        engine.dist(db, None, K, T, out=buf)
    e1.record(); torch.cuda.synchronize()
    res.append("s64_%%d %%.4f" %% (s64, e0.elapsed_time(e1) / reps))
    db.close(); del buf
print(" ".join(res))
''' % ROOT


def main():
    sos = [x for x in sys.argv[1:] if not x.isdigit()]
    rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
    res = {so: [] for so in sos}
    for _ in range(rounds):
        for so in sos:
            o = subprocess.run([sys.executable, "-c", CHILD, so], capture_output=True, text=True)
            if o.returncode:
                print(o.stderr[-2000:])
                sys.exit(1)
            line = o.stdout.strip().split("\n")[-1]
            res[so].append(line)
            print("%-48s %s" % (so, line), flush=True)
    for so in sos:
        vals = {}
        for line in res[so]:
            t = line.split()
            for i in range(0, len(t), 2):
                vals.setdefault(t[i], []).append(float(t[i + 1]))
        print("%-48s" % so, {k: round(sum(v) / len(v), 4) for k, v in vals.items()})


main()
