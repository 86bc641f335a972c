"""Host-array kernel-2 entry points (the poppunk_refine drop-ins on numpy arrays), 5e7 rows, PCIe inclusive,
next to the CPU oracle on this host's threads."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import poppunk_refine
from oracle import oracle
rows = int(os.environ.get("ROWS", "49995000"))
rng = np.random.Generator(np.random.PCG64(5))
d = (rng.random((rows, 2), dtype=np.float32) * 0.3)
poppunk_refine.assignThreshold(d[:1000], 2, 0.1, 0.1)
for rep in range(5):
    y = None      # the previous result is released outside the timed call
    t0 = time.perf_counter(); y = poppunk_refine.assignThreshold(d, 2, 0.1, 0.1); t = time.perf_counter() - t0
    print("assignThreshold host arrays: %.1f ms (%.1f GB/s of 12 B/row)" % (t * 1e3, rows * 12 / t / 1e9))
for rep in range(2):
    e = None
    t0 = time.perf_counter(); e = poppunk_refine.edgeThreshold_array(d, 2, 0.02, 0.02); t = time.perf_counter() - t0
    print("edgeThreshold host arrays: %.1f ms, %d edges" % (t * 1e3, len(e)))
thr = int(os.environ.get("OMP_THREADS", "16"))
for rep in range(2):
    yo = None
    t0 = time.perf_counter(); yo = oracle.assign_threshold(d, 2, 0.1, 0.1, threads=thr); t = time.perf_counter() - t0
    print("CPU oracle assign, %d threads: %.1f ms" % (thr, t * 1e3))
print("equal:", bool(np.array_equal(y, yo)))
