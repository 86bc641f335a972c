"""PCIe-inclusive host-buffer call (pp_sketchlib.query_arrays -> ppk_query): 10k self, breakdown."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import pp_sketchlib, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
n = int(os.environ.get("N", "10000"))
sk, _ = synth.make_sketches(n, K)
pp_sketchlib.query_arrays(sk[:500], None, K, 16, 14, T)          # load the library, create the context
for rep in range(4):
    h = None
    t0 = time.perf_counter()
    h, nf = pp_sketchlib.query_arrays(sk, None, K, 16, 14, T)
    t = time.perf_counter() - t0
    print("rep %d: %.1f ms  %.2f Gpairs/s  (failed %d, checksum %.6f)" % (rep, t * 1e3, h.shape[0] / t / 1e9, nf, float(h[::9973].sum())))
