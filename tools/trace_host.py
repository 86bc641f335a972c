import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from poppunk_amd import _lib, pp_sketchlib, synth, sketchdb
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(10000, K)
e = pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(10000)], K, sk, 16, 14, T, None, random_status="mapped"))
for hp in (1, 2):
    _lib.set_option("host_parts", hp)
    for rep in range(4):
        out = None
        sys.stderr.write("=== host_parts %d rep %d\n" % (hp, rep)); sys.stderr.flush()
        t0 = time.perf_counter()
        out, _ = pp_sketchlib.query_entries(e, None, K, T, devices=[0])
        sys.stderr.write("=== call %.3f ms\n" % ((time.perf_counter() - t0) * 1e3)); sys.stderr.flush()
