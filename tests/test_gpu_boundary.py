"""Kernel 2 parity on a real MI355X: assignThreshold / edgeThreshold / generateTuples and the
fused distance->edge path vs the CPU oracle (bit-exact: float32 compares and index math)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import engine, poppunk_refine, pp_sketchlib, synth

pytestmark = pytest.mark.gpu

KMERS = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)


def grid10():
    # test/test-refine.py:47-50
    x = np.arange(0, 1, 0.1, dtype=np.float32)
    y = np.arange(0, 1, 0.1, dtype=np.float32)
    xv, yv = np.meshgrid(x, y)
    return np.ascontiguousarray(np.hstack((xv.reshape(-1, 1), yv.reshape(-1, 1))), dtype=np.float32)


class _HipRefine:
    assign = staticmethod(lambda d, slope, x, y: poppunk_refine.assignThreshold(d, slope, x, y, 2))
    edges = staticmethod(lambda d, slope, x, y: poppunk_refine.edgeThreshold_array(d, slope, x, y))
    tuples = staticmethod(lambda a, label: poppunk_refine.generateTuples_array(a, label))
    iterate_1d = staticmethod(poppunk_refine.thresholdIterate1D_arrays)
    iterate_2d = staticmethod(poppunk_refine.thresholdIterate2D_arrays)


def test_pinned_by_reference_test_refine():
    """The HIP path against the reference's own pure-Python withinBoundary / iter_tuples
    (test/test-refine.py:10-38; tests/golden/boundary_refine.npz, see tests/refine_golden.py)."""
    import refine_golden
    assert refine_golden.check(_HipRefine) > 70000


def test_known_answers_grid(golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "boundary_known_answers.json")))
    d = grid10()
    for slope in (0, 1, 2):
        a = poppunk_refine.assignThreshold(d, slope, 0.5, 0.5, 2)
        w, o, out = ka["grid10"]["counts_within_online_outside"][str(slope)]
        assert ((a == -1).sum(), (a == 0).sum(), (a == 1).sum()) == (w, o, out)
        assert np.array_equal(a, oracle.assign_threshold(d, slope, 0.5, 0.5))
    a2 = poppunk_refine.assignThreshold(d, 2, 0.5, 0.5)
    assert np.flatnonzero(a2 == 0).tolist() == ka["grid10"]["slope2_online_rows"]


def test_known_answers_tuples(golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "boundary_known_answers.json")))
    assert poppunk_refine.generateTuples([-1] * 10, -1) == \
        [tuple(t) for t in ka["condensed_n5_all_rows"]]
    assert poppunk_refine.generateTuples([-1] * 6, -1, self=False, num_ref=3) == \
        [tuple(t) for t in ka["generate_tuples_nonself_numref3_2queries_all"]]
    assert poppunk_refine.generateTuples([-1] * 3, -1, self=True, num_ref=0, int_offset=10) == \
        [tuple(t) for t in ka["generate_tuples_self_n3_offset10"]]
    # edgeThreshold on an all-within 5-sample matrix gives the same condensed order
    d = np.zeros((10, 2), dtype=np.float32)
    assert poppunk_refine.edgeThreshold(d, 2, 0.5, 0.5) == \
        [tuple(t) for t in ka["condensed_n5_all_rows"]]


@pytest.mark.parametrize("samples", [2, 3, 100, 363])
@pytest.mark.parametrize("slope", [0, 1, 2])
def test_random_matrix_like_test_refine(samples, slope):
    """test/test-refine.py:64-82 with a seeded matrix, compared element for element."""
    rng = np.random.Generator(np.random.PCG64(100 + samples))
    d = rng.random((samples * (samples - 1) // 2, 2)).astype(np.float32)
    a = poppunk_refine.assignThreshold(d, slope, 0.5, 0.5)
    assert np.array_equal(a, oracle.assign_threshold(d, slope, 0.5, 0.5))
    e = poppunk_refine.edgeThreshold_array(d, slope, 0.5, 0.5)
    assert np.array_equal(e, oracle.edge_threshold(d, slope, 0.5, 0.5))
    # assign == -1 -> generateTuples is the exclusive variant
    t = poppunk_refine.generateTuples_array(a, -1)
    assert np.array_equal(t, oracle.generate_tuples(a.astype(np.int32), -1))
    assert np.array_equal(t, oracle.edge_threshold(d, slope, 0.5, 0.5, inclusive=False))
    # python-tuple form and membership, as the reference test checks it
    tl = poppunk_refine.generateTuples([int(x) for x in a], -1)
    assert tl == [tuple(x) for x in t.tolist()]


def test_points_on_and_near_the_line():
    """FMA sensitivity: points on / within 1 ulp of the slope-2 line
    must classify exactly like the un-fused float32 CPU evaluation."""
    rng = np.random.Generator(np.random.PCG64(20260928))
    x_max, y_max = np.float32(0.0123), np.float32(0.217)
    n = 1 << 20
    x = (rng.random(n) * float(x_max)).astype(np.float32)
    y = ((1.0 - x.astype(np.float64) / float(x_max)) * float(y_max)).astype(np.float32)
    jitter = rng.integers(-1, 2, size=n)
    y = np.where(jitter > 0, np.nextafter(y, np.float32(1)), np.where(jitter < 0, np.nextafter(y, np.float32(-1)), y)).astype(np.float32)
    d = np.ascontiguousarray(np.stack([x, y], axis=1))
    a = poppunk_refine.assignThreshold(d, 2, float(x_max), float(y_max))
    want = oracle.assign_threshold(d, 2, float(x_max), float(y_max), threads=4)
    assert np.array_equal(a, want)
    assert (want == 0).sum() > 1000 and (want == -1).sum() > 1000 and (want == 1).sum() > 1000
    # numpy float32 restatement of ((y0*x_max)+(x0*y_max))-(x_max*y_max), un-fused
    s = (y * x_max + x * y_max) - x_max * y_max
    assert np.array_equal(np.sign(s).astype(np.float32), want)


def test_degenerate_boundary_sqrt_branch():
    d = np.asarray([[0, 0], [0.1, 0.2], [0, 0.3]], dtype=np.float32)
    for xm, ym in ((0.0, 0.5), (0.5, 0.0)):
        assert np.array_equal(poppunk_refine.assignThreshold(d, 2, xm, ym),
                              oracle.assign_threshold(d, 2, xm, ym))


def test_generate_tuples_nonself_and_labels():
    rng = np.random.Generator(np.random.PCG64(5))
    num_ref, num_q = 37, 23
    a = rng.integers(0, 4, size=num_ref * num_q).astype(np.int32)
    for label in (0, 3):
        got = poppunk_refine.generateTuples_array(a, label, self=False, num_ref=num_ref, int_offset=5)
        want = oracle.generate_tuples(a, label, self=False, num_ref=num_ref, int_offset=5)
        assert np.array_equal(got, want)
    assert poppunk_refine.generateTuples(a, 9, self=False, num_ref=num_ref) == []


def test_noconvert_type_errors():
    d64 = np.zeros((10, 2), dtype=np.float64)
    with pytest.raises(TypeError):
        poppunk_refine.assignThreshold(d64, 2, 0.5, 0.5)
    with pytest.raises(TypeError):
        poppunk_refine.edgeThreshold(np.zeros((2, 10), dtype=np.float32).T, 2, 0.5, 0.5)
    with pytest.raises(RuntimeError):
        poppunk_refine.edgeThreshold(np.zeros((4, 2), dtype=np.float32), 2, 0.5, 0.5)  # 4 != n(n-1)/2


def test_large_stream_and_edge_order():
    """n = 3000 samples -> 4.5e6 rows: many compaction blocks; edges must come out in row order."""
    n = 3000
    rng = np.random.Generator(np.random.PCG64(1))
    d = rng.random((n * (n - 1) // 2, 2)).astype(np.float32) * np.float32(0.2)
    e = poppunk_refine.edgeThreshold_array(d, 2, 0.05, 0.06)
    want = oracle.edge_threshold(d, 2, 0.05, 0.06)
    assert len(want) > 10000 and np.array_equal(e, want)
    a = poppunk_refine.assignThreshold(d, 2, 0.05, 0.06)
    assert np.array_equal(a, oracle.assign_threshold(d, 2, 0.05, 0.06, threads=4))


@pytest.mark.parametrize("inclusive", [True, False])
@pytest.mark.parametrize("slope", [0, 1, 2])
def test_fused_edges_equal_two_step(slope, inclusive):
    sk, _ = synth.make_sketches(500, KMERS, cluster_size=25)
    tbl = synth.random_match_table(KMERS)
    # the invariant is fused == two-step on the SAME distances, so the two-step input is the
    # GPU matrix (its agreement with the oracle is test_gpu_dist.py's job)
    dist, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    odist, _ = oracle.query(sk, None, KMERS, 16, 14, tbl, threads=4)
    assert np.abs(dist - odist).max() <= 1e-6
    x_max, y_max = synth.boundary_for_quantile(dist, 0.1)
    scale = (np.float32(dist[:, 0].max()), np.float32(dist[:, 1].max()))
    scaled = np.ascontiguousarray(dist / np.asarray(scale, dtype=np.float32))   # models.py:1085
    assert scaled.dtype == np.float32
    xs, ys = x_max / float(scale[0]), y_max / float(scale[1])
    want = oracle.edge_threshold(scaled, slope, xs, ys, inclusive=inclusive)
    db = engine.SketchDB(sk, 16, 14, device=0)
    got, _ = engine.dist_edges(db, None, KMERS, tbl, slope=slope, x_max=xs, y_max=ys, scale=scale,
                               inclusive=inclusive)
    assert len(want) > 100
    assert np.array_equal(got.cpu().numpy(), want)
    # band-split edge lists concatenate to the whole (multi-GPU config 5 shape)
    b = engine.band_split(500, 0, 4)
    parts = [engine.dist_edges(db, None, KMERS, tbl, slope=slope, x_max=xs, y_max=ys, scale=scale,
                               inclusive=inclusive, q_begin=b[i], q_end=b[i + 1], cap=16)[0]
             for i in range(4)]
    import torch
    assert np.array_equal(torch.cat(parts).cpu().numpy(), want)
    db.close()


def test_fused_edges_ref_query():
    sk, _ = synth.make_sketches(400, KMERS, cluster_size=20)
    tbl = synth.random_match_table(KMERS)
    ref, qry = sk[:290], sk[290:]
    dist, _ = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, tbl)
    x_max, y_max = synth.boundary_for_quantile(dist, 0.1)
    want = oracle.edge_threshold(dist, 2, x_max, y_max, n_ref=290, inclusive=False)
    rdb, qdb = engine.SketchDB(ref, 16, 14), engine.SketchDB(qry, 16, 14)
    got, _ = engine.dist_edges(rdb, qdb, KMERS, tbl, slope=2, x_max=x_max, y_max=y_max,
                               inclusive=False)
    assert len(want) > 100 and np.array_equal(got.cpu().numpy(), want)
    # the unfused device path gives the same list
    dd, _ = engine.dist(rdb, qdb, KMERS, tbl)
    e2 = engine.edge_threshold_dev(dd, 2, x_max, y_max, n_ref=290, inclusive=False)
    assert np.array_equal(e2.cpu().numpy(), want)
    a = engine.assign_threshold_dev(dd, 2, x_max, y_max)
    assert np.array_equal(a.cpu().numpy(), oracle.assign_threshold(dist, 2, x_max, y_max))
    rdb.close()
    qdb.close()


# ---- "next" rows: the boundary sweeps ----------------------------------------

@pytest.fixture(params=[1, 0], ids=["bisection", "every-boundary"])
def sweep_window(request):
    """The classify pass of the sweeps both ways: a row's count by bisection over nested boundaries (the default),
    and by evaluating every boundary (option sweep_window 0)."""
    from poppunk_amd import _lib
    old = _lib.get_option("sweep_window")
    _lib.set_option("sweep_window", request.param)
    yield request.param
    _lib.set_option("sweep_window", old)


def test_threshold_iterate_rows_on_and_beside_nested_boundaries(sweep_window):
    """40 nested boundaries (refine's outward sweep); rows planted ON every boundary, one and a few ulps either side of
    it, and at relative distances around the bisection's 2^-20 margin -- where a probe must report "neither" and the
    wavefront takes the full evaluation -- among random rows, negative coordinates included.  (No NaN rows: with a NaN
    key the reference's comparator, boundary.hpp:35-37, is not a strict weak order and its sorted order is undefined.)"""
    rng = np.random.Generator(np.random.PCG64(4242))
    n = 900
    rows = n * (n - 1) // 2
    d = (rng.random((rows, 2)) * 0.5).astype(np.float32)
    x0, y0, x1, y1 = 0.05, 0.06, 0.30, 0.34
    offsets = np.linspace(0.0, float(np.hypot(x1 - x0, y1 - y0)), 40)
    planted = []
    for off in offsets:
        xm, ym = oracle.boundary_of_offset(off, 2, x0, y0, x1, y1)
        xm, ym = np.float32(xm), np.float32(ym)
        for t in rng.random(24):
            x = np.float32(t) * xm
            y = np.float32((1.0 - float(x) / float(xm)) * float(ym))
            for rel in (0.0, 6e-8, -6e-8, 2.4e-7, -2.4e-7, 9.0e-7, -9.0e-7, 9.6e-7, -9.6e-7, 1.1e-6, -1.1e-6, 3e-6, -3e-6):
                planted.append((x, np.float32(float(y) * (1.0 + rel))))
                planted.append((np.nextafter(x, np.float32(1)), np.float32(float(y) * (1.0 + rel))))
    planted = np.asarray(planted, dtype=np.float32)
    where = rng.choice(rows, size=len(planted), replace=False)
    d[where] = planted
    d[rng.choice(rows, 50, replace=False), 0] = np.float32(-0.01)
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(d, offsets, 2, x0, y0, x1, y1)
    wi, wj, wo = oracle.threshold_iterate_1d(d, offsets, 2, x0, y0, x1, y1)
    assert len(wi) > 10000
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)
    # the 2-D sweep over x_max at one y_max is nested too
    x_max = np.linspace(0.08, 0.4, 20).astype(np.float32)
    gi, gj, go = poppunk_refine.thresholdIterate2D_arrays(d, x_max, 0.3)
    wi, wj, wo = oracle.threshold_iterate_2d(d, x_max, 0.3)
    assert len(wi) > 10000
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)


def test_sweep_tickets_are_one_sequence_per_device():
    """The host waits for the sweep's scan kernel by polling a ticket in pinned memory.  The first sweep of a process
    with <= 124 offsets (16-bit packed words) followed by the first with more (32-bit words) are two instantiations of
    the same host code: with a ticket counter each, the second call would wait for a number the pinned word already
    holds and read the control block before its kernel had run.  A fresh process, exactly that order."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from poppunk_amd import poppunk_refine
from oracle import oracle
rng = np.random.Generator(np.random.PCG64(5))
n = 1500
d = (rng.random((n * (n - 1) // 2, 2)) * 0.6).astype(np.float32)
for n_off in (12, 130, 12, 130):
    off = np.linspace(0.0, 0.4, n_off)
    got = poppunk_refine.thresholdIterate1D_arrays(d, off, 2, 0.05, 0.06, 0.3, 0.34)
    want = oracle.threshold_iterate_1d(d, off, 2, 0.05, 0.06, 0.3, 0.34)
    assert len(want[0]) > 100000 and all(np.array_equal(g, w) for g, w in zip(got, want)), n_off
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("slope", [0, 1, 2])
@pytest.mark.parametrize("samples", [3, 100, 700])
def test_threshold_iterate_1d(samples, slope, sweep_window):
    """poppunk_refine.thresholdIterate1D (boundary.cpp:154-210) element for element, plus the
    reference test's own check (test/test-refine.py:84-110): edges with offset index <= o are
    exactly the rows assignThreshold puts within (<= 0) boundary o."""
    rng = np.random.Generator(np.random.PCG64(7 + samples))
    d = (rng.random((samples * (samples - 1) // 2, 2)) * 0.6).astype(np.float32)
    offsets = np.linspace(-0.05, 0.25, 12) * np.sqrt(2)
    x0, y0, x1, y1 = 0.1, 0.12, 0.3, 0.36
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(d, offsets, slope, x0, y0, x1, y1)
    wi, wj, wo = oracle.threshold_iterate_1d(d, offsets, slope, x0, y0, x1, y1)
    assert len(wi) > 0 or samples == 3
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)
    for oi, off in enumerate(offsets):
        xm, ym = oracle.boundary_of_offset(off, slope, x0, y0, x1, y1)
        a = poppunk_refine.assignThreshold(d, slope, xm, ym)
        want = set(map(tuple, oracle.edge_threshold(d, slope, xm, ym).tolist()))
        assert want == {(int(a_), int(b_)) for a_, b_, c_ in zip(gi, gj, go) if c_ <= oi}
        assert len(want) == int((a <= 0).sum())
    # list-returning form, as pybind returns it
    li, lj, lo = poppunk_refine.thresholdIterate1D(d, list(offsets), slope, x0, y0, x1, y1, 2)
    assert li == gi.tolist() and lj == gj.tolist() and lo == go.tolist()


def test_threshold_iterate_1d_edge_cases(sweep_window):
    rng = np.random.Generator(np.random.PCG64(99))
    d = (rng.random((4950, 2)) * 0.5).astype(np.float32)
    with pytest.raises(RuntimeError, match="must be sorted"):
        poppunk_refine.thresholdIterate1D(d, [0.1, 0.0], 2, 0.2, 0.2, 0.3, 0.3)
    # nothing within any boundary
    i, j, o = poppunk_refine.thresholdIterate1D(d + np.float32(5), [0.0, 0.1], 2, 0.2, 0.2, 0.3, 0.3)
    assert i == [] and j == [] and o == []
    # every row within the last boundary (where the reference reads one past the end)
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(d, [0.0, 3.0], 2, 0.2, 0.2, 0.3, 0.3)
    wi, wj, wo = oracle.threshold_iterate_1d(d, [0.0, 3.0], 2, 0.2, 0.2, 0.3, 0.3)
    assert len(gi) == 4950 and np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)
    # a sweep that moves towards the origin (boundaries shrink): the exact sequential path
    off = np.linspace(0.0, 0.2, 5)
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(d, off, 2, 0.3, 0.3, 0.1, 0.1)
    wi, wj, wo = oracle.threshold_iterate_1d(d, off, 2, 0.3, 0.3, 0.1, 0.1)
    assert len(wi) > 0 and np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)
    # duplicate distances: ties must keep row order (stable sort)
    dd = np.repeat(d[:100], 3, axis=0)[:276]      # 24 samples -> 276 rows
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(dd, off, 2, 0.1, 0.1, 0.3, 0.3)
    wi, wj, wo = oracle.threshold_iterate_1d(dd, off, 2, 0.1, 0.1, 0.3, 0.3)
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)


@pytest.mark.parametrize("samples", [3, 100, 700])
def test_threshold_iterate_2d(samples, sweep_window):
    """poppunk_refine.thresholdIterate2D (boundary.cpp:212-237), and test-refine.py:112-138."""
    rng = np.random.Generator(np.random.PCG64(17 + samples))
    d = (rng.random((samples * (samples - 1) // 2, 2)) * 0.6).astype(np.float32)
    x_max = np.asarray([0.1, 0.2, 0.3, 0.45], dtype=np.float32)
    y_max = 0.2
    gi, gj, go = poppunk_refine.thresholdIterate2D_arrays(d, x_max, y_max)
    wi, wj, wo = oracle.threshold_iterate_2d(d, x_max, y_max)
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)
    for oi, xm in enumerate(x_max):
        want = set(map(tuple, oracle.edge_threshold(d, 2, float(xm), y_max).tolist()))
        assert want == {(int(a_), int(b_)) for a_, b_, c_ in zip(gi, gj, go) if c_ <= oi}
    with pytest.raises(RuntimeError, match="must be sorted"):
        poppunk_refine.thresholdIterate2D(d, [0.2, 0.1], 0.2)


def test_threshold_iterate_on_real_distances(sweep_window):
    """The --fit-model refine shape: 40 offsets over a resident distance matrix."""
    sk, _ = synth.make_sketches(3000, KMERS, cluster_size=30)
    tbl = synth.random_match_table(KMERS)
    dist, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    scale = dist.max(axis=0)
    x = np.ascontiguousarray(dist / scale)
    m0 = np.quantile(x, 0.01, axis=0)
    m1 = np.quantile(x, 0.5, axis=0)
    offsets = np.linspace(0.0, float(np.linalg.norm(m1 - m0)), 40)
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(x, offsets, 2, m0[0], m0[1], m1[0], m1[1])
    wi, wj, wo = oracle.threshold_iterate_1d(x, offsets, 2, m0[0], m0[1], m1[0], m1[1])
    assert len(wi) > 10000
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)


def test_qc_edges_on_device():
    """qcDistMat's masks + generateTuples (PopPUNK/qc.py:332-337,:349-354) on the resident matrix."""
    import torch
    rng = np.random.Generator(np.random.PCG64(8))
    n = 400
    d = (rng.random((n * (n - 1) // 2, 2)) * np.asarray([0.06, 0.7])).astype(np.float32)
    d[rng.integers(0, len(d), 500), 0] = 0.0
    d[rng.integers(0, len(d), 300), 1] = 0.0
    dt = torch.from_numpy(d).cuda()
    max_pi, max_a = 0.05, 0.6
    long_rows = np.where((d[:, 0] > np.float32(max_pi)) | (d[:, 1] > np.float32(max_a)), 0, 1)
    want = oracle.generate_tuples(long_rows.astype(np.int32), 0)
    got = engine.qc_edges_dev(dt, max_pi, max_a).cpu().numpy()
    assert len(want) > 100 and np.array_equal(got, want)
    zero_rows = np.where((d[:, 0] == 0) | (d[:, 1] == 0), 0, 1)
    want = oracle.generate_tuples(zero_rows.astype(np.int32), 0)
    got = engine.qc_edges_dev(dt, max_pi, max_a, zero=True).cpu().numpy()
    assert len(want) > 500 and np.array_equal(got, want)
    # ref x query layout (poppunk_assign QC, qc.py:406-407)
    nr = 37
    dq = d[:nr * 50]
    want = oracle.generate_tuples(np.where((dq[:, 0] > np.float32(max_pi)) | (dq[:, 1] > np.float32(max_a)), 0, 1)
                                  .astype(np.int32), 0, self=False, num_ref=nr)
    got = engine.qc_edges_dev(torch.from_numpy(dq).cuda(), max_pi, max_a, n_ref=nr).cpu().numpy()
    assert np.array_equal(got, want)


def test_end_to_end_clusters_recover_the_synthetic_population(tmp_path):
    """Whole path: sketches -> fused distance/boundary -> edge list -> connected components
    (the hand-off network.construct_network_from_edge_list + printClusters consume): planted
    clusters (members share ~96 % of bins, different clusters ~25 %) are recovered by an
    accessory threshold; and the dense matrix survives the .dists.pkl/.npy round trip."""
    from poppunk_amd import distfile
    rng = np.random.Generator(np.random.PCG64(21))
    n, csize, nbins = 600, 20, 1024
    n_clu = n // csize
    species = rng.integers(0, 1 << 14, size=(5, nbins), dtype=np.uint16)
    roots = np.where(rng.random((n_clu, 5, nbins)) < 0.5, species[None],
                     rng.integers(0, 1 << 14, size=(n_clu, 5, nbins), dtype=np.uint16))
    member = np.arange(n) % n_clu
    bins = roots[member].copy()
    redraw = rng.random(bins.shape) < 0.02
    bins[redraw] = rng.integers(0, 1 << 14, size=int(redraw.sum()), dtype=np.uint16)
    sk = synth.bitslice(bins, 14)
    tbl = synth.random_match_table(KMERS)
    db = engine.SketchDB(sk, 16, 14)
    dist, n_failed = engine.dist(db, None, KMERS, tbl)
    d = dist.cpu().numpy()
    assert int(n_failed.item()) == 0
    same = np.asarray([member[i] == member[j] for i in range(n) for j in range(i + 1, n)])
    assert d[same, 1].max() < 0.15 and d[~same, 1].min() > 0.5      # accessory separates them
    edges, _ = engine.dist_edges(db, None, KMERS, tbl, slope=1, x_max=0.0, y_max=0.3, inclusive=False)
    assert len(edges) == int(same.sum())
    n_comp, labels = distfile.clusters_from_edges(n, edges.cpu().numpy())
    assert n_comp == n_clu
    for g in range(n_clu):
        assert len(set(labels[member == g])) == 1
    names = ["s%d" % i for i in range(n)]
    distfile.storePickle(names, names, True, d, str(tmp_path / "x.dists"))
    _, _, self_flag, back = distfile.readPickle(str(tmp_path / "x.dists"), enforce_self=True)
    assert self_flag and np.array_equal(back, d)
    db.close()


def test_nan_and_inf_rows_follow_the_reference_branches():
    """src/boundary.cpp:66-78: NaN is neither == 0 nor > 0, so the else branch gives -1, while
    edge_iterate's `<= 0` is false for NaN (no edge); +inf is outside, -inf within."""
    d = np.asarray([[np.nan, 0.1], [0.1, np.nan], [np.inf, 0.1], [-np.inf, 0.0], [0.0, 0.0],
                    [0.5, 0.0]], dtype=np.float32)                     # 6 rows = 4 samples
    for slope in (0, 1, 2):
        a = poppunk_refine.assignThreshold(d, slope, 0.5, 0.5)
        assert np.array_equal(a, oracle.assign_threshold(d, slope, 0.5, 0.5))
        assert np.array_equal(poppunk_refine.edgeThreshold_array(d, slope, 0.5, 0.5),
                              oracle.edge_threshold(d, slope, 0.5, 0.5))
    a2 = poppunk_refine.assignThreshold(d, 2, 0.5, 0.5)
    assert a2[0] == -1 and a2[1] == -1 and a2[2] == 1 and a2[3] == -1 and a2[4] == -1 and a2[5] == 0
    e = poppunk_refine.edgeThreshold(d, 2, 0.5, 0.5)
    assert (0, 1) not in e and (0, 2) not in e and (1, 2) in e and (2, 3) in e   # rows 0,1 NaN; 3,5 within/on


def test_refine_boundary_model_mirror():
    """models.RefineBoundary = RefineFit.assign / apply_threshold (PopPUNK/models.py:956-994,
    :1065-1091) + the generateTuples hand-off (network.py:1180-1184): host path, resident path and
    the fused sketch -> edge-list path agree with the oracle's statement of the same steps."""
    import torch
    from poppunk_amd import engine, models
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk = synth.make_sketches(420, kmers, cluster_size=35, seed=11)[0]
    X, _ = oracle.query(sk, None, kmers, 16, 14, tbl, threads=4)
    scale = np.amax(X, axis=0)                                 # models.py:253,:861
    xs = X / scale
    x_max, y_max = synth.boundary_for_quantile(xs, 0.15)
    b = models.RefineBoundary(scale=scale, slope=2, optimal_x=x_max, optimal_y=y_max,
                              core_boundary=0.4 * x_max, accessory_boundary=0.4 * y_max)
    for slope, (xm, ym) in ((None, (x_max, y_max)), (0, (0.4 * x_max, 0)), (1, (0, 0.4 * y_max))):
        want = oracle.assign_threshold(xs, 2 if slope is None else slope, xm, ym)
        assert np.array_equal(b.assign(X, slope), want)
        got = b.assign_dev(torch.as_tensor(X, device="cuda"), slope).cpu().numpy()
        assert np.array_equal(got, want)
        edges_want = oracle.generate_tuples(want.astype(np.int32), -1, True, 0, 0)
        assert b.edges(X, slope=slope) == [tuple(e) for e in edges_want.tolist()]
        db = engine.SketchDB(sk, 16, 14)
        fused, _ = b.edges_from_sketches(db, None, kmers, tbl, slope=slope)
        assert np.array_equal(fused.cpu().numpy(), edges_want)
        db.close()
    t = models.RefineBoundary.from_threshold(float(np.quantile(X[:, 0], 0.1)))
    assert t.slope == 0 and np.array_equal(t.scale, [1, 1])
    assert np.array_equal(t.assign(X), oracle.assign_threshold(X, 0, t.core_boundary, 0))
    with pytest.raises(RuntimeError):
        models.RefineBoundary().assign(X)


def test_threshold_iterate_2d_on_resident_matrix():
    import torch
    rng = np.random.Generator(np.random.PCG64(42))
    n = 300
    d = (rng.random((n * (n - 1) // 2, 2)) * 0.4).astype(np.float32)
    x_max = np.sort(rng.uniform(0.02, 0.3, size=20)).astype(np.float32)
    wi, wj, wo = oracle.threshold_iterate_2d(d, x_max, 0.21)
    gi, gj, go = engine.threshold_iterate_2d_dev(torch.as_tensor(d, device="cuda"), x_max, 0.21, cap=4)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gj.cpu().numpy(), wj)
    assert np.array_equal(go.cpu().numpy(), wo)
    with pytest.raises(RuntimeError):
        engine.threshold_iterate_2d_dev(torch.as_tensor(d, device="cuda"), x_max[::-1].copy(), 0.21)


def test_threshold_iterate_many_offsets_and_one_pass_host_calls():
    """More offsets than fit a kernel argument (the boundaries live in device memory: up to 1023 per
    call; the reference has no limit, its callers pass 40 and 20), and the host entry points'
    protocol: a call with too little room leaves the finished result parked on the device and
    reports its size, ppk_parked_fetch copies it out (tests/test_gpu_hostcalls.py has the rest)."""
    import ctypes as C
    from poppunk_amd import _lib
    rng = np.random.Generator(np.random.PCG64(77))
    samples = 260
    d = rng.random((samples * (samples - 1) // 2, 2)).astype(np.float32)
    offsets = np.linspace(-0.25, 0.3, 300) * np.sqrt(2)
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(d, offsets, 2, 0.2, 0.2, 0.3, 0.3)
    wi, wj, wo = oracle.threshold_iterate_1d(d, offsets, 2, 0.2, 0.2, 0.3, 0.3)
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo) and len(wi) > 1000
    xs = np.linspace(0.01, 0.6, 300).astype(np.float32)
    gi, gj, go = poppunk_refine.thresholdIterate2D_arrays(d, xs, 0.2)
    wi, wj, wo = oracle.threshold_iterate_2d(d, xs, 0.2)
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo) and len(wi) > 1000
    with pytest.raises(RuntimeError, match="too many offsets"):
        poppunk_refine.thresholdIterate2D_arrays(d, np.linspace(0.01, 0.6, 1024).astype(np.float32), 0.2)
    # size query + explicit fetch through the raw C ABI
    lib = _lib.lib()
    n = C.c_size_t(0)
    fp = d.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.ppk_edge_threshold(fp, d.shape[0], 0, 2, 0.5, 0.5, 1, 0, None, 0, C.byref(n)) == _lib.ERR_CAPACITY
    want = oracle.edge_threshold(d, 2, 0.5, 0.5)
    assert n.value == len(want) > 0
    ij = np.empty((n.value, 2), dtype=np.int64)
    assert lib.ppk_parked_fetch(ij.ctypes.data_as(C.POINTER(C.c_longlong)), None, None, n.value, None) == _lib.OK
    assert np.array_equal(ij, want)
    # different arguments after a parked result: a fresh computation, not the parked list
    e2 = poppunk_refine.edgeThreshold_array(d, 2, 0.4, 0.5)
    assert np.array_equal(e2, oracle.edge_threshold(d, 2, 0.4, 0.5))


def test_forked_workers_can_use_the_library_when_the_parent_has_not_touched_the_gpu():
    """SURVEY 8(b), threading: refine's 2-D mode calls thresholdIterate2D from forked
    multiprocessing.Pool workers (refine.py:147-163).  Loading libppk_hip.so must not create a HIP
    context (options and devices are set up at first use), so children forked from a parent that has
    only IMPORTED the module each bring up their own."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from poppunk_amd import _lib, poppunk_refine
_lib.lib()                                   # loaded in the parent, GPU untouched
rng = np.random.Generator(np.random.PCG64(3))
d = rng.random((4950, 2), dtype=np.float32)          # 100 samples, condensed
xs = np.linspace(0.2, 0.8, 5).astype(np.float32)
pids = []
for w in range(2):
    pid = os.fork()
    if pid == 0:
        i, j, o = poppunk_refine.thresholdIterate2D_arrays(d, xs, 0.5)
        a0 = poppunk_refine.assignThreshold(d, 2, float(xs[0]), 0.5)
        ok = len(i) == len(j) == len(o) and (o == 0).sum() == (a0 <= 0).sum()
        os._exit(0 if ok else 3)
    pids.append(pid)
bad = [os.waitpid(p, 0)[1] for p in pids]
sys.exit(1 if any(bad) else 0)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_quickstart_example_runs_end_to_end(tmp_path):
    """examples/quickstart.py: .h5 database -> queryDatabase -> dists.pkl/.npy -> boundary -> edges ->
    clusters, and the fused edge list equal to the unfused one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "quickstart.py"), "400", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "identical" in r.stdout


# ---- generateAllTuples (src/boundary.cpp:125-150): the dense network's pair list ------------------

@pytest.mark.parametrize("num_ref,num_queries,self_,off", [
    (2, 0, True, 0), (3, 0, True, 0), (65, 0, True, 0), (1000, 0, True, 17), (2049, 0, True, -3),
    (1, 0, True, 0), (0, 0, True, 0), (3, 5, False, 0), (5, 3, False, 4), (640, 97, False, 0), (7, 0, False, 0),
])
def test_generate_all_tuples(num_ref, num_queries, self_, off):
    want = oracle.generate_all_tuples(num_ref, num_queries, self_, off)
    got = poppunk_refine.generateAllTuples_array(num_ref, num_queries, self_, off)
    assert got.dtype == np.int64 and np.array_equal(got, want.reshape(-1, 2))
    if len(want) < 5000:
        t = poppunk_refine.generateAllTuples(num_ref, num_queries, self=self_, int_offset=off)
        assert isinstance(t, list) and t == [tuple(r) for r in want.tolist()]


def test_generate_all_tuples_full_size_and_device_entry():
    """10 000 samples -> 49 995 000 pairs: first, last, a checksum and sortedness instead of a second copy."""
    import ctypes as C
    import torch
    from poppunk_amd import _lib
    n = 10000
    got = poppunk_refine.generateAllTuples_array(n)
    assert got.shape == (n * (n - 1) // 2, 2)
    assert got[0].tolist() == [0, 1] and got[-1].tolist() == [n - 2, n - 1]
    assert int(got[:, 0].sum()) == sum(i * (n - 1 - i) for i in range(n))
    assert int(got[:, 1].sum()) == sum(j * j for j in range(n))
    key = got[:, 0] * n + got[:, 1]
    assert bool(np.all(np.diff(key) > 0))
    # the device entry point: too little room is an error with the count, nothing written
    buf = torch.zeros((10, 2), dtype=torch.int64, device="cuda:0")
    ne = C.c_size_t(0)
    rc = _lib.lib().ppk_generate_all_tuples_dev(7, 0, 1, 0, C.c_void_p(buf.data_ptr()), 10, C.byref(ne), None)
    assert rc == _lib.ERR_CAPACITY and ne.value == 21 and int(buf.abs().sum()) == 0
    buf = torch.zeros((21, 2), dtype=torch.int64, device="cuda:0")
    rc = _lib.lib().ppk_generate_all_tuples_dev(7, 0, 1, 0, C.c_void_p(buf.data_ptr()), 21, C.byref(ne), None)
    torch.cuda.synchronize()
    assert rc == 0 and np.array_equal(buf.cpu().numpy(), oracle.generate_all_tuples(7))
