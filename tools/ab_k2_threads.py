"""assignThreshold / edgeThreshold on HOST arrays (5e7 rows = 400 MB in, 200 MB out): one call, against
the same rows cut into 2 / 4 pieces handed to the library from as many Python threads at once (ctypes
releases the GIL) -- does the PCIe-bound host path gain from several copies in flight, as ppk_query did?"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import poppunk_refine as R
rng = np.random.Generator(np.random.PCG64(1))
n = 49995000
d = rng.random((n, 2), dtype=np.float32)
R.assignThreshold(d[:1000], 2, 0.5, 0.5)
def run(parts):
    cuts = [n * i // parts // 64 * 64 for i in range(parts)] + [n]
    outs = [None] * parts
    def work(i):
        outs[i] = R.assignThreshold(d[cuts[i]:cuts[i + 1]], 2, 0.5, 0.5)
    ts = []
    for rep in range(6):
        th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[1:])
    return ts[len(ts) // 2], ts[0], float(np.concatenate(outs)[::9973].sum())
for parts in (1, 2, 4):
    med, mn, chk = run(parts)
    print("assignThreshold 5e7 rows, %d concurrent piece(s): median %.2f ms  min %.2f ms  checksum %.1f" % (parts, med, mn, chk), flush=True)
