"""Neighbours from the tiles at 100 000 genomes with and without the staged opening (option knn_warm: the job
opens with 1/knn_warm of its rows, cuts the candidate list -- every bound drops to the k-th distance so far --
and runs the rest under those bounds), k = 5 / 10 / 20; same box, alternating, results compared bit for bit.

    gpurun -- python tools/ab_knn_warm.py [n_genomes]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from poppunk_amd import _lib, engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
db = engine.SketchDB(synth.make_sketches_device(n, kmers, device="cuda:0"), 16, 14, device=0)
for knn in (5, 10, 20):
    ref = None
    for warm in (0, 32, 0, 32):
        _lib.set_option("knn_warm", warm)
        best = None
        for rep in range(3):
            info = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            oi, oj, od = engine.knn_from_sketches(db, kmers, tbl, knn, method="tiles", info=info)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            best = t if best is None or t < best else best
        same = True if ref is None else bool(torch.equal(oj, ref[0]) and torch.equal(od, ref[1]))
        ref = ref or (oj.clone(), od.clone())
        print("n %d k %2d %-22s %7.1f ms (best of 3), list at the end %10d entries, same neighbours: %s"
              % (n, knn, "staged (knn_warm 32)" if warm else "one pass (knn_warm 0)", best * 1e3, info["candidates"], same), flush=True)
_lib.set_option("knn_warm", 32)
