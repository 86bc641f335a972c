"""Kernel 2 streams on 5e7 rows: assign / edges, GB/s."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import engine
rows = 49995000
d = (torch.rand((rows, 2), device="cuda") * 0.3).contiguous()
out = torch.empty(rows, dtype=torch.float32, device="cuda")
def timed(fn, reps=50):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for slope in (2, 0):
    t = timed(lambda: engine.assign_threshold_dev(d, slope, 0.1, 0.1, out=out))
    print("assign slope %d: %.4f ms  %.0f GB/s" % (slope, t * 1e3, rows * 12 / t / 1e9))
t = timed(lambda: engine.assign_threshold_dev(d[1:], 2, 0.1, 0.1, out=out[1:]))
print("assign, unaligned view: %.4f ms  %.0f GB/s" % (t * 1e3, rows * 12 / t / 1e9))
t = timed(lambda: engine.edge_threshold_dev(d, 2, 0.02, 0.02))
print("edges: %.4f ms  %.0f GB/s" % (t * 1e3, rows * 8 / t / 1e9))
