"""k nearest neighbours of N genomes (default 1 000 000) straight from kernel 1's tiles, no distance matrix in
any form (ppk_knn_sketches_dev); sampled samples are checked against a brute-force row of the CPU oracle
(all N distances of that sample, stable order, ties to the lower index -- src/extend.cpp:266-279).

    gpurun -- python tools/scale_knn.py [n_genomes] [knn]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from poppunk_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
knn = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
sk_t = synth.make_sketches_device(n, kmers, device="cuda:0")
db = engine.SketchDB(sk_t, 16, 14, device=0)
for rep in range(2):
    info = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    oi, oj, od = engine.knn_from_sketches(db, kmers, tbl, knn, dist_col=0, method="tiles", info=info)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("%d genomes, %d neighbours each: %.2f s (%.2f G pairs/s over the triangle), %d candidates (%.0f per sample)"
          % (n, knn, t, n * (n - 1) / 2 / t / 1e9, info["candidates"], info["candidates"] / n))
oj, od = oj.cpu().numpy().reshape(n, knn), od.cpu().numpy().reshape(n, knn)
host = sk_t.cpu().numpy().view(np.uint64)
rng = np.random.Generator(np.random.PCG64(8))
bad = 0
rows = rng.choice(n, size=12, replace=False)
for r in rows.tolist():
    d, _ = oracle.query(host, host[r:r + 1], kmers, 16, 14, tbl, threads=16)      # row = q*n_ref + r: all refs against one query
    col = d[:, 0]
    order = np.argsort(col, kind="stable")
    order = order[order != r][:knn]
    if not (np.array_equal(oj[r], order) and np.array_equal(od[r], col[order])):
        bad += 1
print("sampled samples checked against a brute-force oracle row: %d, disagreements: %d" % (len(rows), bad))
