"""thresholdIterate1D (40 offsets) and thresholdIterate2D (20 offsets) on the resident 10 000-genome matrix:
wall time per call (the 1-D call synchronises once inside) and parity with the oracle on the first call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import engine, synth
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10000
K = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32); T = synth.random_match_table(K)
sk, _ = synth.make_sketches(n, K)
db = engine.SketchDB(sk, 16, 14)
d, _ = engine.dist(db, None, K, T)
x = d.cpu().numpy()
scale = torch.tensor([float(x[:, 0].max()), float(x[:, 1].max())], device="cuda")
xs = (d / scale).contiguous()
xh = xs.cpu().numpy()
m0 = np.quantile(xh[::20], 0.01, axis=0)
m1 = np.quantile(xh[::20], 0.30, axis=0)
offs = np.linspace(0.0, float(np.linalg.norm(m1 - m0)), 40)


import ctypes
from poppunk_amd import _lib


try:      # (an older build of the library, tools/ab_sweeps_builds.py: no stage timers)
    _lib.lib().ppk_prof_stages_enable
    HAVE_STAGES = True
except AttributeError:
    HAVE_STAGES = False


def stages(reset=True):
    if not HAVE_STAGES:
        return []
    buf = ctypes.create_string_buffer(8192)
    _lib.lib().ppk_prof_stages_read(buf, 8192, 1 if reset else 0)
    rows = [l.split("\t") for l in buf.value.decode().splitlines()]
    return [(n, float(ms), int(c)) for n, ms, c in rows]


def timed(fn, reps=20):
    out = fn()
    torch.cuda.synchronize()
    ts = []
    if HAVE_STAGES:
        _lib.lib().ppk_prof_stages_enable(0 if "--plain" in sys.argv else 1)      # (--plain: wall time without the stage events)
    stages()
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    if HAVE_STAGES:
        _lib.lib().ppk_prof_stages_enable(0)
    for n, ms, c in stages():
        print("      stage %-16s %8.1f us  (x%d)" % (n, ms / max(c, 1) * 1e3, c))
    ts.sort()
    return out, ts[len(ts) // 2], ts[0]


ti = engine.threshold_iterate_1d_dev(xs, offs, 2, m0[0], m0[1], m1[0], m1[1])
cap = len(ti[0]) + 16
out, med, best = timed(lambda: engine.threshold_iterate_1d_dev(xs, offs, 2, m0[0], m0[1], m1[0], m1[1], cap=cap))
print("1D rows %d offsets %d emitted %d: median %.3f ms  best %.3f ms" % (xs.shape[0], len(offs), out[0].shape[0], med, best))
if "--check" in sys.argv:
    wi, wj, wo = oracle.threshold_iterate_1d(xh, offs, 2, m0[0], m0[1], m1[0], m1[1])
    ok = (np.array_equal(out[0].cpu().numpy(), wi) and np.array_equal(out[1].cpu().numpy(), wj)
          and np.array_equal(out[2].cpu().numpy(), wo))
    print("1D equals the oracle element for element:", ok, len(wi))
    assert ok

xm = np.linspace(float(m0[0]), float(m1[0]) * 1.5, 20).astype(np.float32)
ym = float(m1[1]) * 1.5
t2 = engine.threshold_iterate_2d_dev(xs, xm, ym)
cap2 = len(t2[0]) + 16
out2, med2, best2 = timed(lambda: engine.threshold_iterate_2d_dev(xs, xm, ym, cap=cap2))
print("2D rows %d offsets %d emitted %d: median %.3f ms  best %.3f ms" % (xs.shape[0], len(xm), out2[0].shape[0], med2, best2))
if "--check" in sys.argv:
    wi, wj, wo = oracle.threshold_iterate_2d(xh, xm, ym)
    ok = (np.array_equal(out2[0].cpu().numpy(), wi) and np.array_equal(out2[1].cpu().numpy(), wj)
          and np.array_equal(out2[2].cpu().numpy(), wo))
    print("2D equals the oracle element for element:", ok, len(wi))
    assert ok
