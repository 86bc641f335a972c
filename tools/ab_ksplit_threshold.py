"""Where the one-launch small-job path stops paying: n genomes self and Q queries x 10 000 refs through the tile kernel
(ksplit 0) and through the k-split path (ksplit large), kernel time by the library's HIP events.  The `ksplit` default
is the tile count (at 5 k) below which the k-split path is taken.

    python tools/ab_ksplit_threshold.py > gpurun_out/ksplit_threshold.txt
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import _lib, engine, synth  # noqa: E402

K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
T = synth.random_match_table(K)
lib = _lib.lib()
sk, _ = synth.make_sketches(10000, K)
db10 = engine.SketchDB(sk, 16, 14)
o = torch.empty((25000000, 2), dtype=torch.float32, device="cuda")
nf = torch.zeros(1, dtype=torch.int64, device="cuda")


def kernel_us(fn, reps=60):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        fn()
        torch.cuda.synchronize()
    lib.ppk_prof_enable(1)
    lib.ppk_prof_read(None, None, 1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    lib.ppk_prof_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0)
    lib.ppk_prof_read(C.byref(ms), C.byref(n), 1)
    return ms.value / reps * 1e3


def tiles(n_ref, n_qry):
    rt, qt = (n_ref + 255) // 256, ((n_qry or n_ref) + 31) // 32
    return rt * qt // 2 + qt if not n_qry else rt * qt


print("%-28s %6s %12s %12s %12s" % ("job", "tiles", "tile kernel", "k-split", "k-split S=1"))
SIZES = [int(x) for x in os.environ.get("SIZES", "1200,1600,1800,2000,2200,2500,2800,3200,3600,4000").split(",")]
QS = [int(x) for x in os.environ.get("QS", "64,128,192,256,320,400,512,640").split(",") if x]
jobs = [("%d self" % n, n, 0) for n in SIZES] + [("%d queries x 10 000" % q, 10000, q) for q in QS]
for name, n, q in jobs:
    ref = db10 if q else engine.SketchDB(sk[:n], 16, 14)
    qry = engine.SketchDB(sk[:q], 16, 14) if q else None
    rows = engine.rows_in_band(n, q, 0, q or n)
    fn = lambda: engine.dist(ref, qry, K, T, out=o[:rows], n_failed=nf)
    res = []
    for ks, sl in ((0, 0), (100000, 0), (100000, 1)):
        _lib.set_option("ksplit", ks)
        _lib.set_option("ksplit_slices", sl)
        res.append(kernel_us(fn))
    print("%-28s %6d %10.1f us %10.1f us %10.1f us" % (name, tiles(n, q), res[0], res[1], res[2]))
    if qry is not None:
        qry.close()
    if not q:
        ref.close()
