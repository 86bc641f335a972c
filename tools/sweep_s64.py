"""Kernel-1 rate against the sketch size (12 000 genomes, 5 k), as (s = 1024)-equivalent pairs/s."""
import os, sys, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()
def kms(fn, reps=4):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return ms.value / max(n.value, 1)
K = np.asarray([13, 17, 21, 25, 29], dtype=np.int32); T = synth.random_match_table(K)
n = 12000
for s64 in (16, 24, 32, 48, 64, 96, 156):
    sk = synth.make_sketches_device(n, K, sketchsize64=s64, bbits=14)
    db = engine.SketchDB(sk, s64, 14)
    out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.float32, device="cuda")
    t = kms(lambda: engine.dist(db, None, K, T, out=out))
    pairs = n * (n - 1) // 2
    eq = pairs * s64 / 16
    print("s64=%3d  ref tile = %5.1f MB  db = %5.0f MB  %8.2f ms  %.2f G (s=1024)-equivalent pairs/s" %
          (s64, 256 * 5 * s64 * 14 * 8 / 1e6, n * 5 * s64 * 14 * 8 / 1e6, t, eq / t / 1e6), flush=True)
    del out, sk; db.close(); torch.cuda.empty_cache()
