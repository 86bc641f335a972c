"""One device call on a band with more pair tiles than a dispatch holds (400 000 genomes self: 9.8 M tiles) against
the fused host call, which works through the band in pieces of its own: the lists must be identical.

    gpurun -- python tools/check_400k.py
"""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import engine, synth
n = 400000
kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
tbl = synth.random_match_table(kmers)
sk_t = synth.make_sketches_device(n, kmers, device="cuda:0")
db = engine.SketchDB(sk_t, 16, 14, device=0)
sub = engine.SketchDB(synth.make_sketches_device(2000, kmers, device="cuda:0"), 16, 14, device=0)
d_sub, _ = engine.dist(sub, None, kmers, tbl)
x_max, y_max = synth.boundary_for_quantile(d_sub.cpu().numpy(), 0.02)
t0 = time.perf_counter(); e1, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=32 << 20); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
t0 = time.perf_counter(); e2, _ = engine.edges_host(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=32 << 20); t2 = time.perf_counter() - t0
print("400000 genomes: one device call (several launches) %.2f s, host call in pieces %.2f s, %d edges, identical: %s"
      % (t1, t2, len(e2), bool(np.array_equal(e1.cpu().numpy(), e2))))
