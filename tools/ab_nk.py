"""Throughput against the number of k-mer lengths at s=1024 (count register of 2 dwords up to 5 lengths, 3 up to 8, 4 beyond)."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
lib = _lib.lib()
def kms(fn, reps=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:      # clock ramp
        fn(); torch.cuda.synchronize()
    lib.ppk_prof_enable(1); lib.ppk_prof_read(None, None, 1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0); lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return ms.value / max(n.value, 1)
n = int(os.environ.get("N", "10000"))
pairs = n * (n - 1) // 2
for kmers in ([13, 21, 29], [13, 17, 21, 25, 29], [13, 16, 19, 22, 25, 28], [13, 15, 17, 19, 21, 23, 25],
              [13, 15, 17, 19, 21, 23, 25, 27, 29]):
    K = np.asarray(kmers, dtype=np.int32); T = synth.random_match_table(K)
    sk, _ = synth.make_sketches(n, K)
    db = engine.SketchDB(sk, 16, 14)
    o = torch.empty((pairs, 2), dtype=torch.float32, device="cuda")
    t = kms(lambda: engine.dist(db, None, K, T, out=o))
    print("nk=%d: %.3f ms  %.2f Gpairs/s  = %.2f G (nk=5)-equivalent pairs/s  [%s]" %
          (len(kmers), t, pairs / t / 1e6, pairs * len(kmers) / 5 / t / 1e6, lib.ppk_last_kernel_name().decode()))
    del db, o
