"""Checker for tests/golden/boundary_refine.npz -- the reference's own pure-Python statement of
kernel 2 (withinBoundary / iter_tuples, test/test-refine.py:10-38, executed by
tests/golden/make_golden.py) -- shared by the oracle test (CPU) and the HIP test (GPU).

`impl` supplies the implementation under test:
    impl.assign(dist, slope, x_max, y_max)                      -> float32 [n]   (-1 / 0 / +1)
    impl.edges(dist, slope, x_max, y_max)                       -> int64 [m, 2]  (line_dist <= 0)
    impl.tuples(assign, within_label)                           -> int64 [m, 2]  (assign == label)
    impl.iterate_1d(dist, offsets, slope, x0, y0, x1, y1)       -> (i, j, offset_idx)
    impl.iterate_2d(dist, x_max, y_max)                         -> (i, j, offset_idx)

What the reference test asserts, and how the fixture is used:
  * grid (test-refine.py:47-61): assignThreshold == withinBoundary, all rows (check_res);
  * random matrix (:64-82): every withinBoundary/iter_tuples pair is in generateTuples' list and
    every edgeThreshold pair is in iter_tuples' list (check_tuples) -- here element for element,
    in row order;
  * sweeps (:84-138): per offset, the set of pairs emitted up to that offset equals
    {assign <= 0} at that offset's boundary.
withinBoundary calls a row "on the line" when |in_tri| < float32 eps, src/boundary.cpp when it is
exactly 0.  Rows inside that band but not exactly zero are marked `band` in the fixture; for those
(only) the expected value is the sign of the un-fused float32 expression.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return np.load(os.path.join(HERE, "golden", "boundary_refine.npz"))


def _pairs(samples):
    i, j = np.triu_indices(samples, k=1)
    return np.stack([i, j], axis=1).astype(np.int64)


def _sign_f32(d, slope, x_max, y_max):
    xm, ym = np.float32(x_max), np.float32(y_max)
    if slope == 2:
        t = (d[:, 1] * xm + d[:, 0] * ym) - xm * ym          # float32, un-fused
    elif slope == 0:
        t = d[:, 0] - xm
    else:
        t = d[:, 1] - ym
    return np.sign(t).astype(np.float32)


def check(impl):
    g = load()
    n_checked = 0
    # ---- grid: equality on every row (the fixture has no band rows here) ---------------------
    grid = np.ascontiguousarray(g["grid"], dtype=np.float32)
    for slope in (0, 1, 2):
        assert not g["grid_band%d" % slope].any()
        a = impl.assign(grid, slope, 0.5, 0.5)
        assert a.dtype == np.float32
        assert np.array_equal(a, g["grid_assign%d" % slope]), "grid, slope %d" % slope
        n_checked += len(a)
    # ---- seeded random matrices ---------------------------------------------------------------
    for samples in (100, 363):
        d = np.ascontiguousarray(g["rand%d" % samples], dtype=np.float32)
        pairs = _pairs(samples)
        for slope in (0, 1, 2):
            want = g["rand%d_assign%d" % (samples, slope)].copy()
            band = g["rand%d_band%d" % (samples, slope)]
            a = impl.assign(d, slope, 0.5, 0.5)
            assert np.array_equal(a[~band], want[~band]), "rand%d slope %d" % (samples, slope)
            # the three exact on-line rows planted by make_golden.py are in the `== 0` class
            assert (want[~band] == 0).sum() >= 1
            # band rows: withinBoundary says 0 (eps band), the C++ says the exact sign
            want[band] = _sign_f32(d, slope, 0.5, 0.5)[band]
            assert np.array_equal(a, want)
            # iter_tuples(withinBoundary == -1) == generateTuples(assign, -1), element for element
            ref_edges = g["rand%d_edges%d" % (samples, slope)]
            band_pairs = {tuple(p) for p in pairs[band].tolist()}
            got = impl.tuples(a, -1)
            assert [tuple(p) for p in got.tolist() if tuple(p) not in band_pairs] == \
                [tuple(p) for p in ref_edges.tolist() if tuple(p) not in band_pairs]
            assert np.array_equal(got, pairs[want == -1])
            # edgeThreshold is `<= 0`: the same list plus the on-line rows, in row order
            assert np.array_equal(impl.edges(d, slope, 0.5, 0.5), pairs[want <= 0])
            n_checked += len(a)
    # ---- sweeps (100-sample matrix) -----------------------------------------------------------
    d = np.ascontiguousarray(g["rand100"], dtype=np.float32)
    pairs = _pairs(100)
    x0, y0, x1, y1 = (float(v) for v in g["it1d_line"])
    i, j, o = impl.iterate_1d(d, g["it1d_offsets"], 2, x0, y0, x1, y1)
    i, j, o = np.asarray(i), np.asarray(j), np.asarray(o)
    assert np.all(np.diff(o) >= 0)
    for oi in range(len(g["it1d_offsets"])):
        band = {tuple(p) for p in pairs[g["it1d_band%d" % oi]].tolist()}
        want = {tuple(p) for p in pairs[g["it1d_rows%d" % oi]].tolist()} - band
        got = {(int(a_), int(b_)) for a_, b_, c_ in zip(i, j, o) if c_ <= oi} - band
        assert got == want, "thresholdIterate1D offset %d" % oi
    i, j, o = impl.iterate_2d(d, g["it2d_xmax"].astype(np.float32), float(g["it2d_ymax"]))
    i, j, o = np.asarray(i), np.asarray(j), np.asarray(o)
    for oi in range(len(g["it2d_xmax"])):
        band = {tuple(p) for p in pairs[g["it2d_band%d" % oi]].tolist()}
        want = {tuple(p) for p in pairs[g["it2d_rows%d" % oi]].tolist()} - band
        got = {(int(a_), int(b_)) for a_, b_, c_ in zip(i, j, o) if c_ <= oi} - band
        assert got == want, "thresholdIterate2D offset %d" % oi
    return n_checked
