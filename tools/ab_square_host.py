"""Host-array longToSquare / squareToLong (pp_sketchlib drop-ins), 10k samples, PCIe inclusive."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import pp_sketchlib
n = int(os.environ.get("N", "10000"))
v = np.random.Generator(np.random.PCG64(1)).random(n * (n - 1) // 2, dtype=np.float32)
pp_sketchlib.longToSquare(v[:45].reshape(-1, 1))
for rep in range(3):
    t0 = time.perf_counter(); sq = pp_sketchlib.longToSquare(v.reshape(-1, 1)); t = time.perf_counter() - t0
    print("longToSquare %d: %.1f ms" % (n, t * 1e3))
for rep in range(3):
    t0 = time.perf_counter(); lg = pp_sketchlib.squareToLong(sq); t = time.perf_counter() - t0
    print("squareToLong %d: %.1f ms" % (n, t * 1e3))
print("round trip equal:", bool(np.array_equal(lg.ravel(), v)))
