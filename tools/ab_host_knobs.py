"""The host call (10k self, mirror path: resident handles) against a few knobs, median / min of 10 calls."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import _lib, pp_sketchlib, synth, sketchdb
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
n = int(os.environ.get("N", "10000"))
sk, _ = synth.make_sketches(n, K)
e = pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(n)], K, sk, 16, 14, T, None, random_status="mapped"))
lib = _lib.lib()
def run(label, **opts):
    for k, v in opts.items():
        _lib.set_option(k, v)
    ts, dp = [], []
    for rep in range(12):
        out = None
        t0 = time.perf_counter()
        out, _ = pp_sketchlib.query_entries(e, None, K, T, devices=[0])
        ts.append((time.perf_counter() - t0) * 1e3)
        st = (C.c_double * 7)(); lib.ppk_query_last_stats(st, 7); dp.append(st[4])
    ts = sorted(ts[2:]); dp = sorted(dp[2:])
    print("%-46s call median %.2f min %.2f | device phase median %.2f min %.2f ms" % (label, ts[len(ts) // 2], ts[0], dp[len(dp) // 2], dp[0]), flush=True)
base = dict(host_parts=2, prefault_threads=8, chunk_rows=8 << 20)
run("default (2 parts, 8 touchers, 8 Mi rows)", **base)
for pt in (4, 12, 16, 24):
    run("prefault_threads %d" % pt, **dict(base, prefault_threads=pt))
for cr in (2 << 20, 4 << 20, 16 << 20):
    run("chunk_rows %d Mi" % (cr >> 20), **dict(base, chunk_rows=cr))
run("1 part, 16 touchers", host_parts=1, prefault_threads=16, chunk_rows=8 << 20)
run("3 parts, 16 touchers, 4 Mi", host_parts=3, prefault_threads=16, chunk_rows=4 << 20)
run("2 parts, 16 touchers, 4 Mi", host_parts=2, prefault_threads=16, chunk_rows=4 << 20)
run("default again", **base)
