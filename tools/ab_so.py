"""Same-box A/B of two builds of libppk_hip.so: the kernel-1 time of the 10k self job (and a
10 240^2 ref x query job), alternating A, B, A, B ... in separate processes so that clock / thermal
drift and box-to-box differences cancel.

    python tools/ab_so.py A.so B.so [C.so ...] [rounds]

`A.so@PPK_ABLATE=32` runs that build with the variable set (the library reads PPK_* once at load).
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from poppunk_amd import _lib
_lib.SO_PATH = os.path.abspath(sys.argv[1])
import torch
# an older build lacks the entry points added since: bind what it has (the jobs below use ppk_dist_dev only)
import ctypes
_lib._preload_hip_runtime()
_h = ctypes.CDLL(_lib.SO_PATH)
for _n in list(_lib.SIGNATURES):
    if not hasattr(_h, _n):
        del _lib.SIGNATURES[_n]
from poppunk_amd import engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
n = int(os.environ.get("N", "10000"))
sk, _ = synth.make_sketches(max(n, 10240), K)
out = {}
for name, (nr, nq) in (("self%%d" %% n, (n, 0)), ("rq10240", (10240, 10240)), ("c4_50000x10000", (10000, 50000))):
    ref = engine.SketchDB(sk[:nr], 16, 14)
    qry = engine.SketchDB(np.concatenate([sk[:10000]] * ((nq + 9999) // 10000))[:nq], 16, 14) if nq else None
    buf = None
    for _ in range(40 if nq <= 10240 else 5):
        buf, _f = engine.dist(ref, qry, K, T, out=buf)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 60 if nq <= 10240 else 10
    ev0.record()
    for _ in range(reps):
        engine.dist(ref, qry, K, T, out=buf)
    ev1.record(); torch.cuda.synchronize()
    out[name] = ev0.elapsed_time(ev1) / reps
    del buf
print(" ".join("%%s %%.4f" %% kv for kv in out.items()))
''' % ROOT

def main():
    sos = [x for x in sys.argv[1:] if not x.isdigit()]
    rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
    res = {so: [] for so in sos}
    for r in range(rounds):
        for so in sos:
            path, _, assign = so.partition("@")
            env = dict(os.environ)
            if assign:
                env.update([assign.split("=", 1)])
            o = subprocess.run([sys.executable, "-c", CHILD, path], capture_output=True, text=True, env=env)
            if o.returncode:
                print(o.stderr[-2000:]); sys.exit(1)
            line = o.stdout.strip().split("\n")[-1]
            res[so].append(line)
            print("%-44s %s" % (so, line), flush=True)
    for so in sos:
        vals = {}
        for line in res[so]:
            t = line.split()
            for i in range(0, len(t), 2):
                vals.setdefault(t[i], []).append(float(t[i + 1]))
        print("%-44s" % so, {k: round(sum(v) / len(v), 4) for k, v in vals.items()})

main()
