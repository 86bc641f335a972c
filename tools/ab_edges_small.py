"""Small fused distance -> boundary -> edge-list jobs at the BASELINE sketch size (s = 1 024): tile kernel ("ksplit" 0)
against the k-split path (round 5: the tile's last unit applies the boundary).  Wall time of engine.dist_edges
including its count read-back; same lists.   python tools/ab_edges_small.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
allsk = synth.make_sketches_device(11000, K, device="cuda:0")
ref10k = engine.SketchDB(allsk[:10000].contiguous(), 16, 14)
d, _ = engine.dist(ref10k, None, K, T, q_begin=0, q_end=300)
x_max, y_max = synth.boundary_for_quantile(synth.tensor_to_numpy(d), 0.02)
jobs = [("%d self" % n, engine.SketchDB(allsk[:n].contiguous(), 16, 14), None) for n in (500, 1000, 2000, 3000, 4000)]
jobs += [("%d queries x 10 000 refs" % q, ref10k, engine.SketchDB(allsk[10000:10000 + q].contiguous(), 16, 14)) for q in (1, 32, 100, 400, 1000)]
for name, ref, qry in jobs:
    res, cnt = [], []
    for ks in (0, 1200):
        _lib.set_option("ksplit", ks)
        for _ in range(5): e, _ = engine.dist_edges(ref, qry, K, T, slope=2, x_max=x_max, y_max=y_max, cap=1 << 22)
        ts = []
        for _ in range(30):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e, _ = engine.dist_edges(ref, qry, K, T, slope=2, x_max=x_max, y_max=y_max, cap=1 << 22)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        res.append(sorted(ts)[len(ts) // 2]); cnt.append(len(e))
    assert cnt[0] == cnt[1]
    print("%-28s edges=%7d  tile kernel %8.1f us   k-split %8.1f us" % (name, cnt[0], res[0], res[1]), flush=True)
_lib.set_option("ksplit", 1200)
