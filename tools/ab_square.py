"""10240 x 10240 ref x query job (the shape tools/ubench_pipe.hip runs) under PPK_ABLATE."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _exp; _exp.use()      # ablate / map / edge_list_keep: the experiments build
from poppunk_amd import _lib, engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
lib = _lib.lib()
def kms(fn, reps=int(os.environ.get("REPS", "5"))):
    for _ in range(int(os.environ.get("WARM", "0"))): fn()
    fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return ms.value / max(n.value, 1)
n = int(os.environ.get("N", "10240"))
sk, _ = synth.make_sketches(n, K)
if os.environ.get("ZERO"):
    sk[:] = 0x5a5a5a5a5a5a5a5a
a = engine.SketchDB(sk, 16, 14); b = a if os.environ.get('SAME') else engine.SketchDB(sk.copy(), 16, 14)
o = torch.empty((n * n, 2), dtype=torch.float32, device="cuda")
t = kms(lambda: engine.dist(a, b, K, T, out=o))
print("SAME=%s " % os.environ.get("SAME","-") + "N=%d ZERO=%s ABLATE=%s: %.3f ms %.2f Gpairs/s" % (n, os.environ.get("ZERO", "-"), os.environ.get("PPK_ABLATE", "-"), t, n * n / t / 1e6))
