"""Default PopPUNK sketch size (s = 9984 -> sketchsize64 156), 6 k-mer lengths: the 128-bit packed path."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()
import time
def kms(fn, reps=5):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:      # clock ramp
        fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return ms.value / max(n.value, 1)
for kmers in ([13, 17, 21, 25, 29], [13, 16, 19, 22, 25, 28], [13, 17, 21, 25])[:1 if os.environ.get('ONLY5') else 3]:
    K = np.asarray(kmers, dtype=np.int32); T = synth.random_match_table(K)
    n = int(os.environ.get('N', '4000'))
    sk, _ = synth.make_sketches(n, K, sketchsize64=156, bbits=14, cluster_size=50)
    db = engine.SketchDB(sk, 156, 14)
    t = kms(lambda: engine.dist(db, None, K, T))
    pairs = n * (n - 1) // 2
    # work per pair scales with nk * sketchsize64: express as equivalent s=1024/nk=5 pairs
    eq = pairs * (len(kmers) * 156) / (5 * 16)
    print("s=9984 nk=%d: %.2f ms, %.3f Gpairs/s, = %.2f G (s=1024,nk=5)-equivalent pairs/s  [%s]" % (len(kmers), t, pairs / t / 1e6, eq / t / 1e6, lib.ppk_last_kernel_name().decode()))
