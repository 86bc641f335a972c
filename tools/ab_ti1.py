"""thresholdIterate1D on the resident 10k matrix: per-kernel breakdown under rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(10000, K)
db = engine.SketchDB(sk, 16, 14)
d, _ = engine.dist(db, None, K, T)
x = d.cpu().numpy()
scale = torch.tensor([float(x[:, 0].max()), float(x[:, 1].max())], device="cuda")
xs = (d / scale).contiguous()
xh = xs.cpu().numpy()
m0 = np.quantile(xh[::20], 0.01, axis=0)
m1 = np.quantile(xh[::20], 0.30, axis=0)
offs = np.linspace(0.0, float(np.linalg.norm(m1 - m0)), 40)
ti = engine.threshold_iterate_1d_dev(xs, offs, 2, m0[0], m0[1], m1[0], m1[1])
for _ in range(5):
    out = engine.threshold_iterate_1d_dev(xs, offs, 2, m0[0], m0[1], m1[0], m1[1], cap=len(ti[0]) + 16)
torch.cuda.synchronize()
print("emitted", out[0].shape[0])
