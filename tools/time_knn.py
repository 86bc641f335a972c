"""Neighbours from the tiles (10 per sample, 10 000 genomes): wall time per call, median of 12."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
knn = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(n, K)
db = engine.SketchDB(sk, 16, 14)
ts = []
info = {}
for _ in range(14):
    t0 = time.perf_counter(); r = engine.knn_from_sketches(db, K, T, knn, method="tiles", info=info); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[2:])
print("n %d knn %d: median %.3f ms  best %.3f ms  candidates %s" % (n, knn, ts[len(ts) // 2], ts[0], info.get("candidates")))
