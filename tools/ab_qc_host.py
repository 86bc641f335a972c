"""qcDistMat's two edge lists on a 10 000-genome matrix (5e7 rows, host numpy array): one upload + two
device passes (poppunk_amd.qc.qc_edge_lists) next to the construction the reference uses -- numpy masks as
0/1 rows, `.tolist()`, then generateTuples over the Python list (PopPUNK/qc.py:331-337,:348-354); the
generateTuples half is timed with this package's host entry point on the int32 array (the reference's
pybind conversion of a 5e7-element Python list comes on top of that)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import poppunk_refine, qc
rows = int(os.environ.get("ROWS", "49995000"))
rng = np.random.Generator(np.random.PCG64(5))
d = np.stack([rng.uniform(0, 0.04, rows).astype(np.float32), rng.uniform(0, 0.6, rows).astype(np.float32)], axis=1)
d[rng.choice(rows, rows // 5000, replace=False), 0] = 0.0
qc.qc_edge_lists(d[:499500], 0, 0.0399, 0.599)
for rep in range(4):
    t0 = time.perf_counter(); a, z = qc.qc_edge_lists(d, 0, 0.0399, 0.599); t = time.perf_counter() - t0
    print("qc_edge_lists (device, one upload): %.1f ms, %d long + %d zero edges" % (t * 1e3, len(a), len(z)))
t0 = time.perf_counter()
long_rows = np.where([(d[:, 0] > 0.0399) | (d[:, 1] > 0.599)], 0, 1)[0]
zero_rows = np.where([(d[:, 0] == 0) | (d[:, 1] == 0)], 0, 1)[0]
t_mask = time.perf_counter() - t0
t0 = time.perf_counter(); ll = long_rows.tolist(); zl = zero_rows.tolist(); t_list = time.perf_counter() - t0
t0 = time.perf_counter()
a2 = poppunk_refine.generateTuples_array(long_rows.astype(np.int32), 0, self=True)
z2 = poppunk_refine.generateTuples_array(zero_rows.astype(np.int32), 0, self=True)
t_gen = time.perf_counter() - t0
print("reference construction: numpy masks %.0f ms + .tolist() %.0f ms + generateTuples on the arrays %.0f ms"
      % (t_mask * 1e3, t_list * 1e3, t_gen * 1e3))
print("equal:", bool(np.array_equal(a, a2) and np.array_equal(z, z2)))
