"""PCIe-inclusive host-buffer call (pp_sketchlib.query_arrays -> ppk_query): 10k self, fresh result
array each call, against the sub-band size (option chunk_rows) and the database cache."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import _lib, pp_sketchlib, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
n = int(os.environ.get("N", "10000"))
sk, _ = synth.make_sketches(n, K)
pp_sketchlib.query_arrays(sk[:500], None, K, 16, 14, T)          # load the library, create the context
for cache in (1, 0):
    _lib.set_option("db_cache", cache)
    for chunk in (32 << 20, 16 << 20, 8 << 20, 4 << 20, 2 << 20):
        _lib.set_option("chunk_rows", chunk)
        ts = []
        for rep in range(6):
            h = None
            t0 = time.perf_counter()
            h, nf = pp_sketchlib.query_arrays(sk, None, K, 16, 14, T)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[1:])
        print("db_cache %d chunk_rows %3d Mi: median %.2f ms  min %.2f ms  %.2f Gpairs/s  (checksum %.6f)"
              % (cache, chunk >> 20, ts[len(ts) // 2], ts[0], h.shape[0] / ts[len(ts) // 2] / 1e6, float(h[::9973].sum())))
