import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from poppunk_amd import _lib, pp_sketchlib, sketchdb, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(13000, K)
mk = lambda a: pp_sketchlib._Entry(sketchdb.LoadedSketches(["g%d" % i for i in range(len(a))], K, a, 16, 14, T, None, random_status="mapped"))
r = mk(sk[:10000])
for nq in (100, 300, 600, 1000, 1500, 2000, 3000):
    q = mk(sk[10000:10000+nq])
    ts=[]
    for i in range(8):
        t0=time.perf_counter(); out,_=pp_sketchlib.query_entries(r,q,K,T,devices=[0]); ts.append((time.perf_counter()-t0)*1e3); del out
    print("%5d queries x 10000: %8d rows  %s ms" % (nq, nq*10000, " ".join("%.2f"%t for t in ts)), flush=True)
    if nq == 1000:
        _lib.set_option("host_trace", 1)
        out,_=pp_sketchlib.query_entries(r,q,K,T,devices=[0]); del out
        _lib.set_option("host_trace", 0)
    q.close()
