"""C4 shape (50k queries x 10k refs) under ablations / tile orders; prints kernel ms."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _exp; _exp.use()      # ablate / map / edge_list_keep: the experiments build
from poppunk_amd import _lib, engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
lib = _lib.lib()
def kms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return ms.value / max(n.value, 1)
nq = int(os.environ.get("NQ", "50000"))
sk, _ = synth.make_sketches(10000, K)
# queries: tile the 10k set to the requested size (content does not matter for timing)
skq = np.concatenate([sk] * ((nq + 9999) // 10000))[:nq]
db10 = engine.SketchDB(sk, 16, 14); dbq = engine.SketchDB(skq, 16, 14)
o = torch.empty((nq * 10000, 2), dtype=torch.float32, device="cuda")
t = kms(lambda: engine.dist(db10, dbq, K, T, out=o))
print("NQ=%d MAP=%s ABLATE=%s: %.2f ms %.2f Gpairs/s" % (nq, os.environ.get("PPK_MAP", "-"), os.environ.get("PPK_ABLATE", "-"), t, nq * 10000 / t / 1e6))
