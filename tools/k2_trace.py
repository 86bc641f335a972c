"""Kernel 2's edge route on the 10 000-genome matrix, four copies in rotation (cold Infinity Cache), many repetitions:
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  python tools/k2_trace.py [reps]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
db = engine.SketchDB(synth.make_sketches_device(10000, K, device="cuda:0"), 16, 14)
d, _ = engine.dist(db, None, K, T)
sample = synth.tensor_to_numpy(d[torch.randint(0, d.shape[0], (200000,), device=d.device)])
x_max, y_max = synth.boundary_for_quantile(sample, 0.02)
mats = [d] + [d.clone() for _ in range(3)]
n = d.shape[0]
out = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn in (("assign", lambda m: engine.assign_threshold_dev(m, 2, x_max, y_max, out=out)),
                 ("edges", lambda m: engine.edge_threshold_dev(m, 2, x_max, y_max, cap=1 << 22))):
    for i in range(8): fn(mats[i % 4])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(mats[i % 4]); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    print("%s: min %.4f median %.4f max %.4f ms" % (name, t[0], t[len(t) // 2], t[-1]))
