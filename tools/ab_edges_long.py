"""Fused distance -> boundary -> edge list on long sketches (s = 9 984): tile kernel ("ksplit_long" 0) against the
k-split path.   python tools/ab_edges_long.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
for kmers in (np.arange(13, 30, 4), np.arange(6, 16)):
    kmers = kmers.astype(np.int32)
    tbl = synth.random_match_table(kmers, genome_length=20000 if kmers[0] < 10 else 2000000)
    allsk = synth.make_sketches_device(10000, kmers, sketchsize64=156, seed=3, device="cuda:0", chunk=512)
    for n in (1000, 3000, 10000):
        db = engine.SketchDB(allsk[:n].contiguous(), 156, 14)
        d, _ = engine.dist(db, None, kmers, tbl, q_begin=0, q_end=min(n, 200))
        x_max, y_max = synth.boundary_for_quantile(synth.tensor_to_numpy(d), 0.02)
        res, cnt = [], []
        for long_ in (0, 1):
            _lib.set_option("ksplit_long", long_)
            for _ in range(2): e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=1 << 22)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=1 << 22)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 3 * 1e3); cnt.append(len(e))
        assert cnt[0] == cnt[1]
        print("nk=%2d n=%5d edges=%d  tile kernel %8.3f ms   k-split %8.3f ms" % (len(kmers), n, cnt[0], res[0], res[1]), flush=True)
        db.close()
    del allsk
_lib.set_option("ksplit_long", 1)
