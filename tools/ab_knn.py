"""k nearest neighbours straight from the sketches: tiles (MODE_KNN) vs square vs bands; time and
peak HBM (torch's allocator + the library's scratch, from hipMemGetInfo)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
for n in (10000, 50000, 100000):
    t = synth.make_sketches_device(n, K, seed=n, device="cuda:0")
    db = engine.SketchDB(t, 16, 14); del t
    for method in ("tiles", "square", "bands"):
        if method == "square" and n > 46340:
            continue
        if method == "bands" and n > 50000:
            continue
        torch.cuda.empty_cache(); _lib.lib().ppk_release_scratch(); torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        info = {}
        low = [free0]
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = engine.knn_from_sketches(db, K, T, 5, method=method, info=info)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            low.append(torch.cuda.mem_get_info()[0])
        print("n %6d %-6s %9.2f ms  extra HBM held after the call %.2f GB  candidates %s"
              % (n, method, dt * 1e3, (free0 - min(low)) / 1e9, info.get("candidates", "-")), flush=True)
        del r
    db.close()
