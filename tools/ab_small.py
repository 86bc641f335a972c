"""Small jobs (1 000 self; a few queries against 10 000 refs): k-split counts + regression vs the fused tile kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import engine, synth
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(10000, K)
db10 = engine.SketchDB(sk, 16, 14); db1 = engine.SketchDB(sk[:1000], 16, 14)
dbq = {nq: engine.SketchDB(sk[5000:5000 + nq], 16, 14) for nq in (1, 8, 64)}
def timed(fn, reps=200):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1: fn(); torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
o = torch.empty((13000000, 2), dtype=torch.float32, device="cuda")
print("PPK_KSPLIT=%s" % os.environ.get("PPK_KSPLIT", "default"))
print("  1000 self           : %.1f us" % (timed(lambda: engine.dist(db1, None, K, T, out=o[:499500])) * 1e6))
for nq, d in dbq.items():
    print("  %2d queries x 10k refs: %.1f us" % (nq, timed(lambda: engine.dist(db10, d, K, T, out=o[:10000 * nq])) * 1e6))
for n in (1500, 2000, 2500, 3000, 3500, 4000, 5000):
    d = engine.SketchDB(sk[:n], 16, 14)
    print("  %d self : %.1f us" % (n, timed(lambda: engine.dist(d, None, K, T, out=o[:n * (n - 1) // 2]), reps=100) * 1e6))
