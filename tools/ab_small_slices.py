import os, subprocess, sys
ROOT = "/root/repo"
CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, "%s")
from poppunk_amd import _lib
_lib.SO_PATH = os.path.abspath(sys.argv[1])
import torch
from poppunk_amd import engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(1500, K)
out = []
for n in (200, 300, 500, 700, 1000, 1500):
    db = engine.SketchDB(sk[:n], 16, 14)
    buf = None
    for _ in range(50): buf, _f = engine.dist(db, None, K, T, out=buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): engine.dist(db, None, K, T, out=buf)
    e1.record(); torch.cuda.synchronize()
    out.append("n%%d %%.4f" %% (n, e0.elapsed_time(e1) / 300))
print(" ".join(out))
''' % ROOT
for r in range(3):
    for so in sys.argv[1:]:
        o = subprocess.run([sys.executable, "-c", CHILD, so], capture_output=True, text=True)
        print(os.path.basename(so), o.stdout.strip().split("\n")[-1] if o.returncode == 0 else o.stderr[-500:], flush=True)
