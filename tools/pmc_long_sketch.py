"""One launch each of the 10 000-genome job at PopPUNK's default sketch size through the tile kernel ("ksplit" 0) and
through the k-split path: run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc TCC_HIT_sum TCC_MISS_sum` to compare what the
L2s fetch from the fabric.   python tools/pmc_long_sketch.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
db = engine.SketchDB(synth.make_sketches_device(10000, K, sketchsize64=156, device="cuda:0", chunk=2048), 156, 14)
out = torch.empty((10000 * 9999 // 2, 2), dtype=torch.float32, device="cuda")
for ks in (0, 1200, 0, 1200):
    _lib.set_option("ksplit", ks)
    engine.dist(db, None, K, T, out=out)
    torch.cuda.synchronize()
    print(ks, _lib.lib().ppk_last_kernel_name().decode())
