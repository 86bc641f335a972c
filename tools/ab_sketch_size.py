"""Where the default sketch size (s = 9 984, sketchsize64 = 156) loses its 5 % against s = 1 024: the same job at both
sizes with parts of the kernel switched off (option `ablate`: 1 epilogue, 4 the LDS-DMA of the next block, 8 barriers).

    python tools/ab_sketch_size.py [n_genomes]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _exp; _exp.use()      # ablate / map / edge_list_keep: the experiments build
from poppunk_amd import _lib, engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
for s64 in (16, 156):
    db = engine.SketchDB(synth.make_sketches_device(n, kmers, sketchsize64=s64, device="cuda:0"), s64, 14, device=0)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.float32, device="cuda:0")
    nf = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    ops = len(kmers) * s64 * 30
    for ab in (0, 4, 1, 5, 8, 0):
        _lib.set_option("ablate", ab)
        reps = 12 if s64 == 16 else 3
        for _ in range(2):
            engine.dist(db, None, kmers, tbl, out=out, n_failed=nf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            engine.dist(db, None, kmers, tbl, out=out, n_failed=nf)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("s64=%3d  ablate=%d  %8.3f ms  %.3f of the VALU roof  (%.1f us per tile-block at 512 resident workgroups)"
              % (s64, ab, ms, ops * pairs / (ms * 1e-3) / 78.64e12, ms * 1e3 / (pairs / 8192.0 * len(kmers) * s64 / 512.0)))
    _lib.set_option("ablate", 0)
    db.close()
    del out
