"""A cold process's FIRST host call (10k self through the mirror's path) under the host_trace timeline."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PPK_HOST_TRACE"] = "1"
from poppunk_amd import _lib, pp_sketchlib, synth, sketchdb
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(10000, K)
import torch; torch.cuda.init(); torch.zeros(1, device="cuda")        # the HIP context exists (any GPU library pays that)
e = pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(10000)], K, sk, 16, 14, T, None, random_status="mapped"))
t0 = time.perf_counter()
hs = e.resident([0], None)
t1 = time.perf_counter()
sys.stderr.write("=== ppk_db_create (upload + re-layout): %.2f ms\n" % ((t1 - t0) * 1e3))
out, _ = pp_sketchlib.query_entries(e, None, K, T, devices=[0])
t2 = time.perf_counter()
sys.stderr.write("=== first query on the resident database: %.2f ms\n" % ((t2 - t1) * 1e3))
out = None
t3 = time.perf_counter()
out, _ = pp_sketchlib.query_entries(e, None, K, T, devices=[0])
sys.stderr.write("=== second query: %.2f ms\n" % ((time.perf_counter() - t3) * 1e3))
