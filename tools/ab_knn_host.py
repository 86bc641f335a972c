"""The neighbour host call (ppk_query_knn: sketches in a host array -> (i, j, dist) host arrays) at 100 000 genomes,
one device entry and two (the same GPU listed twice: the bands run one after the other, the host merge is real),
next to the device-resident form.

    gpurun -- python tools/ab_knn_host.py [n_genomes] [knn]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from poppunk_amd import engine, pp_sketchlib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
knn = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
sk_t = synth.make_sketches_device(n, kmers, device="cuda:0")
sk = sk_t.cpu().numpy().view(np.uint64)
db = engine.SketchDB(sk_t, 16, 14, device=0)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    oi, oj, od = engine.knn_from_sketches(db, kmers, tbl, knn, method="tiles")
    torch.cuda.synchronize()
    print("device-resident (ppk_knn_sketches_dev):        %.1f ms" % ((time.perf_counter() - t0) * 1e3))
want = oj.cpu().numpy()
db.close()
del sk_t
for devices in ((0,), (0, 0)):
    for rep in range(3):
        t0 = time.perf_counter()
        i, j, d = pp_sketchlib.query_knn_arrays(sk, kmers, 16, 14, knn, 0, tbl, devices=devices)
        t = time.perf_counter() - t0
        print("host call, %d device entr%s (ppk_query_knn):      %.1f ms%s  identical: %s"
              % (len(devices), "y" if len(devices) == 1 else "ies", t * 1e3,
                 " (first: hashes and uploads the %d MB of sketches)" % (sk.nbytes >> 20) if rep == 0 else "",
                 bool(np.array_equal(j, want))))
