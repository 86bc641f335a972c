import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from poppunk_amd import engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
db = engine.SketchDB(synth.make_sketches_device(10000, K, device="cuda:0"), 16, 14)
for _ in range(6):
    engine.knn_from_sketches(db, K, T, 10, method="tiles")
torch.cuda.synchronize()
