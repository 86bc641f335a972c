"""update_distance_matrices at 10 000 refs + 2 000 queries: the reference's construction on this engine (two host
column slices per matrix, longToSquareMulti per column: six uploads) against the engine's own statement
(pp_sketchlib.squareMatrices -> ppk_long_to_square2: three uploads, columns read in place).  Identical matrices.

    python tools/ab_update_db.py > gpurun_out/ab_update_db.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import pp_sketchlib  # noqa: E402
from poppunk_amd.utils import update_distance_matrices  # noqa: E402

n_ref, n_qry = 10000, 2000
rng = np.random.Generator(np.random.PCG64(1))
rr = rng.random((n_ref * (n_ref - 1) // 2, 2), dtype=np.float32)
qr = rng.random((n_ref * n_qry, 2), dtype=np.float32)
qq = rng.random((n_qry * (n_qry - 1) // 2, 2), dtype=np.float32)
refs = ["r%d" % i for i in range(n_ref)]
qrys = ["q%d" % i for i in range(n_qry)]


def reference_way():
    core = pp_sketchlib.longToSquareMulti(distVec=rr[:, [0]], query_ref_distVec=qr[:, [0]],
                                          query_query_distVec=qq[:, [0]], num_threads=1)
    acc = pp_sketchlib.longToSquareMulti(distVec=rr[:, [1]], query_ref_distVec=qr[:, [1]],
                                         query_query_distVec=qq[:, [1]], num_threads=1)
    return core, acc


def engine_way():
    _, core, acc = update_distance_matrices(refs, rr, qrys, qr, qq)
    return core, acc


a = reference_way()
b = engine_way()
assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
for name, fn in (("column slices + longToSquareMulti x 2 (PopPUNK/utils.py:398-405 on this engine)", reference_way),
                 ("update_distance_matrices -> ppk_long_to_square2", engine_way),
                 ("column slices + longToSquareMulti x 2", reference_way),
                 ("update_distance_matrices -> ppk_long_to_square2", engine_way)):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-90s min %.1f median %.1f max %.1f ms" % (name, min(ts), sorted(ts)[2], max(ts)))
print("12 000 samples: two 576 MB squares out, 576 MB of long matrices in; identical matrices")
