import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from poppunk_amd import poppunk_refine as R, _lib
rng = np.random.Generator(np.random.PCG64(1))
d = rng.random((49995000, 2), dtype=np.float32)
R.assignThreshold(d[:1000], 2, 0.5, 0.5)
for rep in range(3):
    _lib.set_option("host_trace", 1 if rep == 2 else 0)
    y = None
    t0 = time.perf_counter(); y = R.assignThreshold(d, 2, 0.5, 0.5); print("call %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
