"""Throughput of the self job (distances resident, HIP-event kernel time) against n."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import _lib, engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
lib = _lib.lib()
sk = synth.make_sketches_device(max(int(x) for x in os.environ.get("SIZES", "100000").split(",")), K, seed=13)      # drawn on the device: numpy needs minutes for 100k
SIZES = [int(x) for x in os.environ.get("SIZES", "500,1000,2000,3000,5000,10000,14000,20000,30000,50000,100000").split(",")]
for n in SIZES:
    db = engine.SketchDB(sk[:n], 16, 14)
    rows = n * (n - 1) // 2
    out = torch.empty((rows, 2), dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        engine.dist(db, None, K, T, out=out); torch.cuda.synchronize()
    reps = max(2, min(50, int(0.5 / max(rows / 16e9, 1e-5))))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): engine.dist(db, None, K, T, out=out)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    print("n=%6d  pairs=%11d  %9.3f ms  %6.2f Gpairs/s  (%s)" % (n, rows, t * 1e3, rows / t / 1e9, lib.ppk_last_kernel_name().decode()))
    del out; db.close(); torch.cuda.empty_cache()
