#!/usr/bin/env python3
"""Randomised campaign of the boundary sweeps, device against the CPU oracle, element for element:
thresholdIterate1D (slopes 0 / 1 / 2, outward and inward sweeps, 1 .. 130 offsets -- beyond 124 the packed
(row in unit | first boundary) word is 32 bits wide --, boundaries through the data, beside it and on an axis) and
thresholdIterate2D, on matrices of 1 .. 1 500 samples: uniform, clumped near the origin, with blocks of equal rows
(ties keep row order), zeros, negative coordinates, and rows planted on and a few ulps beside the boundaries.
Every case runs with the classify pass's bisection and without it (option sweep_window).

    gpurun -- python tools/soak_sweeps.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from oracle import oracle  # noqa: E402
from poppunk_amd import _lib, poppunk_refine  # noqa: E402


def matrix(rng, rows):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        d = rng.random((rows, 2)) * rng.choice([0.3, 0.6, 1.0])
    elif kind == 1:      # clumped near the origin, a tail
        d = np.abs(rng.normal(0.0, 0.05, (rows, 2))) + (rng.random((rows, 2)) < 0.1) * rng.random((rows, 2)) * 0.5
    elif kind == 2:      # few distinct rows: ties in d0
        base = rng.random((max(2, rows // 50), 2)) * 0.5
        d = base[rng.integers(0, len(base), rows)]
    elif kind == 3:      # two clusters, like within / between strain
        c = rng.random(rows) < 0.2
        d = np.where(c[:, None], rng.normal(0.02, 0.01, (rows, 2)), rng.normal(0.3, 0.08, (rows, 2)))
    else:
        d = rng.random((rows, 2)) * 0.5
        d[rng.random(rows) < 0.05] = 0.0
    d = d.astype(np.float32)
    if rng.random() < 0.3:
        d[rng.random(rows) < 0.01, int(rng.integers(0, 2))] = np.float32(-0.01)
    return np.ascontiguousarray(d)


def plant(rng, d, bounds):
    """rows on / beside the boundaries (x_max, y_max), slope 2"""
    rows = d.shape[0]
    pts = []
    for xm, ym in bounds:
        if not (xm > 0 and ym > 0):
            continue
        for t in rng.random(6):
            x = np.float32(t) * np.float32(xm)
            y = np.float32((1.0 - float(x) / float(xm)) * float(ym))
            for rel in (0.0, 6e-8, -6e-8, 9.5e-7, -9.5e-7, 2e-6, -2e-6):
                pts.append((x, np.float32(float(y) * (1.0 + rel))))
    if not pts:
        return
    pts = np.asarray(pts, dtype=np.float32)
    k = min(len(pts), rows)
    d[rng.choice(rows, k, replace=False)] = pts[:k]


def case(rng):
    n = int(rng.choice([2, 3, 5, int(rng.integers(6, 200)), int(rng.integers(200, 700)), int(rng.integers(700, 1500))]))
    rows = n * (n - 1) // 2
    d = matrix(rng, rows)
    msgs = []
    two_d = rng.random() < 0.35
    if two_d:
        n_off = int(rng.choice([1, 2, 5, 20, 40, 61, 64, 65, 126, 130]))
        lo, hi = sorted(rng.random(2) * 0.6 + 0.01)
        x_max = np.sort((lo + (hi - lo) * rng.random(n_off))).astype(np.float32)
        if rng.random() < 0.1:
            x_max[0] = 0.0
        y_max = float(rng.random() * 0.6 + 0.01) if rng.random() > 0.05 else 0.0
        if rng.random() < 0.5:
            plant(rng, d, [(float(x), y_max) for x in x_max])
        want = oracle.threshold_iterate_2d(d, x_max, y_max)
        desc = "2D n=%4d offsets=%3d y_max=%.3f" % (n, n_off, y_max)
        run = lambda: poppunk_refine.thresholdIterate2D_arrays(d, x_max, y_max)
    else:
        slope = int(rng.choice([0, 1, 2, 2, 2]))
        n_off = int(rng.choice([1, 2, 12, 40, 40, 125, 130]))
        x0, y0 = rng.random(2) * 0.2 + 0.01
        x1, y1 = (x0, y0) + rng.random(2) * 0.4 + 0.02
        if rng.random() < 0.15:      # inward: the boundaries shrink, the closed form does not apply
            x0, y0, x1, y1 = x1, y1, x0, y0
        length = float(np.hypot(x1 - x0, y1 - y0))
        start = float(rng.choice([0.0, -0.1, 0.05]))
        offsets = np.sort(start + rng.random(n_off) * length * rng.choice([0.5, 1.0, 1.5])) if rng.random() < 0.5 \
            else np.linspace(start, start + length, n_off)
        if slope == 2 and rng.random() < 0.6:
            plant(rng, d, [oracle.boundary_of_offset(o, 2, x0, y0, x1, y1) for o in offsets[:: max(1, n_off // 12)]])
        want = oracle.threshold_iterate_1d(d, offsets, slope, x0, y0, x1, y1)
        desc = "1D n=%4d slope=%d offsets=%3d %s" % (n, slope, n_off, "inward" if x1 < x0 else "outward")
        run = lambda: poppunk_refine.thresholdIterate1D_arrays(d, offsets, slope, x0, y0, x1, y1)
    for w in (1, 0):
        _lib.set_option("sweep_window", w)
        got = run()
        if not all(np.array_equal(g, x) for g, x in zip(got, want)):
            msgs.append("sweep_window %d: %d rows listed, oracle %d" % (w, len(got[0]), len(want[0])))
    _lib.set_option("sweep_window", 1)
    return desc + " listed=%d" % len(want[0]), msgs


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.Generator(np.random.PCG64(seed))
    bad = 0
    t0 = time.time()
    for c in range(n_cases):
        desc, msgs = case(rng)
        bad += bool(msgs)
        print("case %4d %s  %s" % (c, desc, "ok" if not msgs else "MISMATCH: " + "; ".join(msgs)), flush=True)
    print("%d cases (each with and without the bisection), %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
