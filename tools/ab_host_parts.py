"""PCIe-inclusive host call (10k self, fresh result array per call) against the number of worker
threads that share ONE GPU (the same device listed 1, 2, 3, 4 times: each entry has its own compute /
copy streams and sub-band buffers) and the sub-band size.  Does a second download thread keep the
link busier than one?"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import _lib, pp_sketchlib, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
n = int(os.environ.get("N", "10000"))
sk, _ = synth.make_sketches(n, K)
pp_sketchlib.query_arrays(sk[:500], None, K, 16, 14, T)
lib = _lib.lib()
for chunk in (8 << 20, 4 << 20, 2 << 20):
    _lib.set_option("chunk_rows", chunk)
    _lib.set_option("host_parts", 1)      # the entries are listed explicitly here
    for devs in ((0,), (0, 0), (0, 0, 0), (0, 0, 0, 0)):
        ts = []
        dp = []
        for rep in range(10):
            h = None                      # the previous result is released OUTSIDE the timed call
            t0 = time.perf_counter()
            h, nf = pp_sketchlib.query_arrays(sk, None, K, 16, 14, T, devices=devs)
            ts.append((time.perf_counter() - t0) * 1e3)
            st = (C.c_double * 7)(); lib.ppk_query_last_stats(st, 7)
            dp.append(st[4])
        ts = sorted(ts[2:]); dp = sorted(dp[2:])
        print("chunk_rows %2d Mi  entries %d: call median %.2f ms  min %.2f ms | device phase median %.2f min %.2f ms, downloads in flight %d  checksum %.6f"
              % (chunk >> 20, len(devs), ts[len(ts) // 2], ts[0], dp[len(dp) // 2], dp[0], int(st[2]), float(h[::9973].sum())), flush=True)
