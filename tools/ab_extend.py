"""Queries joining a lineage model's neighbour matrix (poppunk_refine.extend), 100 000 references with 10 neighbours
each + 2 000 queries, as PopPUNK does it -- queryDatabase (query x ref), queryDatabase (query self), longToSquare,
extend on the two dense matrices (PopPUNK/models.py:1355-1372) -- against ppk_extend_sketches, which takes the
sketches in their place.  Same result, PCIe included on both sides.

    gpurun -- python tools/ab_extend.py [n_ref] [n_qry] [knn]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from poppunk_amd import engine, poppunk_refine, pp_sketchlib, synth  # noqa: E402

n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_qry = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
knn = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
sk_t = synth.make_sketches_device(n_ref + n_qry, kmers, device="cuda:0")
rdb = engine.SketchDB(sk_t[:n_ref].contiguous(), 16, 14, device=0)
qdb = engine.SketchDB(sk_t[n_ref:].contiguous(), 16, 14, device=0)
ref_sk = sk_t[:n_ref].cpu().numpy().view(np.uint64)
qry_sk = sk_t[n_ref:].cpu().numpy().view(np.uint64)
del sk_t
i, j, d = (x.cpu().numpy() for x in engine.knn_from_sketches(rdb, kmers, tbl, knn, method="tiles"))
rr = (i, j, d)
for rep in range(3):
    t0 = time.perf_counter()
    qr, _ = pp_sketchlib.query_arrays(ref_sk, qry_sk, kmers, 16, 14, tbl)
    qq, _ = pp_sketchlib.query_arrays(qry_sk, None, kmers, 16, 14, tbl)
    t1 = time.perf_counter()
    qq_sq = pp_sketchlib.longToSquare(np.ascontiguousarray(qq[:, 0]))
    qr_rect = np.ascontiguousarray(qr[:, 0].reshape(n_qry, n_ref).T)
    t2 = time.perf_counter()
    dense = poppunk_refine.extend_arrays(rr, qq_sq, qr_rect, knn)
    t3 = time.perf_counter()
    print("dense route: distances %.1f ms (%.2f GB to the host) + reshape %.1f ms + extend %.1f ms = %.1f ms"
          % ((t1 - t0) * 1e3, (qr.nbytes + qq.nbytes) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
for rep in range(3):
    t0 = time.perf_counter()
    fused = engine.extend_from_sketches(rr, rdb, qdb, kmers, tbl, knn)
    t = time.perf_counter() - t0
    print("from the sketches (ppk_extend_sketches): %.1f ms   identical: %s"
          % (t * 1e3, bool(all(np.array_equal(a, b) for a, b in zip(fused, dense)))))
