"""Full-size parity: 10 000 self on the GPU against the CPU oracle, every one of the 49 995 000 rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle
from poppunk_amd import engine, pp_sketchlib, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
for related in (True, False):
    sk, _ = synth.make_sketches(10000, K, related=related)
    db = engine.SketchDB(sk, 16, 14)
    g, gf = engine.dist(db, None, K, T)
    g = g.cpu().numpy()
    t0 = time.time(); w, wf = oracle.query(sk, None, K, 16, 14, T, threads=16); t = time.time() - t0
    d = np.abs(g - w)
    c, _ = pp_sketchlib.query_arrays(sk[:3000], None, K, 16, 14, counts=True)
    ce = np.array_equal(c, oracle.match_counts(sk[:3000], None, 16, 14, threads=16))
    print("related=%s: failed fits gpu %d / oracle %d; max |d core| %.3g, max |d acc| %.3g; rows not bit-identical: %d of %d; "
          "counts (3000 self) bit-identical: %s; oracle %.1f s" % (related, gf, wf, d[:, 0].max(), d[:, 1].max(),
          int((g != w).any(axis=1).sum()), len(g), ce, t))
    db.close()
