"""A/B of tile orders (PPK_MAP=0/1) on self and ref x query shapes; prints kernel ms."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _exp; _exp.use()      # ablate / map / edge_list_keep: the experiments build
from poppunk_amd import _lib, engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
lib = _lib.lib()
def kms(fn, reps=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:      # clock ramp
        fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return ms.value / max(n.value, 1)
sk, _ = synth.make_sketches(30000, K)
db30 = engine.SketchDB(sk, 16, 14); db13 = engine.SketchDB(sk[:13000], 16, 14)
sk = sk[:20000]
db10 = engine.SketchDB(sk[:10000], 16, 14); dbq = engine.SketchDB(sk[10000:], 16, 14); db20 = engine.SketchDB(sk, 16, 14)
o = torch.empty((450000000, 2), dtype=torch.float32, device="cuda")
for name, fn, pairs in (("self 30k", lambda: engine.dist(db30, None, K, T, out=o[:449985000]), 449985000),
                        ("self 13k", lambda: engine.dist(db13, None, K, T, out=o[:84493500]), 84493500),
                        ("self 10k", lambda: engine.dist(db10, None, K, T, out=o[:49995000]), 49995000),
                        ("self 20k", lambda: engine.dist(db20, None, K, T, out=o[:199990000]), 199990000),
                        ("10k refs x 10k qry", lambda: engine.dist(db10, dbq, K, T, out=o[:100000000]), 100000000),
                        ("10k refs x 2k qry", lambda: engine.dist(db10, dbq, K, T, q_end=2000, out=o[:20000000]), 20000000),
                        ("20k refs x 10k qry(first 10k of same)", lambda: engine.dist(db20, db10, K, T, out=o[:200000000]), 200000000)):
    t = kms(fn)
    print("PPK_MAP=%s %-40s %.3f ms  %.2f Gpairs/s" % (os.environ.get("PPK_MAP", "default"), name, t, pairs / t / 1e6))
