"""Randomised shapes on a real MI355X: ragged sample counts, random bands, random boundaries --
every case compared with the CPU oracle (counts bit-exact, distances <= 1e-6, kernel 2 exact)."""
import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import engine, poppunk_refine, pp_sketchlib, synth

pytestmark = pytest.mark.gpu
KMERS = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
TBL = synth.random_match_table(KMERS)


@pytest.fixture(scope="module")
def pool():
    return synth.make_sketches(900, KMERS, cluster_size=45, seed=77)[0]


@pytest.mark.parametrize("seed", range(12))
def test_random_self_and_refquery_shapes(pool, seed):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    n = int(rng.integers(2, 700))
    idx = rng.choice(900, size=n, replace=False)
    sk = pool[idx]
    counts, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, counts=True)
    assert np.array_equal(counts, oracle.match_counts(sk, None, 16, 14, threads=4))
    got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, TBL)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, TBL, threads=4)
    assert gf == wf and np.abs(got - want).max(initial=0) <= 1e-6
    nr = int(rng.integers(1, n)) if n > 1 else 1
    ref, qry = sk[:nr], sk[nr:]
    if len(qry):
        got, gf = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, TBL)
        want, wf = oracle.query(ref, qry, KMERS, 16, 14, TBL, threads=4)
        assert gf == wf and np.abs(got - want).max(initial=0) <= 1e-6


@pytest.mark.parametrize("seed", range(8))
def test_random_bands_and_fused_edges(pool, seed):
    import torch
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    n = int(rng.integers(40, 700))
    sk = pool[rng.choice(900, size=n, replace=False)]
    db = engine.SketchDB(sk, 16, 14)
    whole, _ = engine.dist(db, None, KMERS, TBL)
    w = whole.cpu().numpy()
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=4)]))
    pieces = [engine.dist(db, None, KMERS, TBL, q_begin=a, q_end=b)[0] for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(torch.cat(pieces), whole)
    slope = int(rng.integers(0, 3))
    x_max, y_max = synth.boundary_for_quantile(w, float(rng.uniform(0.02, 0.6)))
    inclusive = bool(rng.integers(0, 2))
    want = oracle.edge_threshold(w, slope, x_max, y_max, inclusive=inclusive)
    parts = [engine.dist_edges(db, None, KMERS, TBL, slope=slope, x_max=x_max, y_max=y_max,
                               inclusive=inclusive, q_begin=a, q_end=b, cap=8)[0]
             for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(torch.cat(parts).cpu().numpy(), want)
    a = poppunk_refine.assignThreshold(w, slope, x_max, y_max)
    assert np.array_equal(a, oracle.assign_threshold(w, slope, x_max, y_max))
    db.close()


@pytest.mark.parametrize("seed", range(4))
def test_many_ref_tiles_random_bands(seed):
    """9-13 ref tiles (256 samples each) with a ragged right edge and random query bands: every
    workgroup-to-tile decode regime of dist_kernel_v2 (ref tiles below the band, the triangle's
    linear part, saturated ref tiles, strip tiles) -- whole job against the oracle, bands against
    the whole job, ref x query against the oracle."""
    import torch
    rng = np.random.Generator(np.random.PCG64(4000 + seed))
    n = int(rng.integers(2100, 3300))
    sk = synth.make_sketches(n, KMERS, cluster_size=60, seed=500 + seed)[0]
    db = engine.SketchDB(sk, 16, 14)
    whole, gf = engine.dist(db, None, KMERS, TBL)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, TBL, threads=8)
    assert gf == wf and np.abs(whole.cpu().numpy() - want).max() <= 1e-6
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=5)]))
    pieces = [engine.dist(db, None, KMERS, TBL, q_begin=a, q_end=b)[0] for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(torch.cat(pieces), whole)
    nr = int(rng.integers(1200, n - 300))
    dbr, dbq = engine.SketchDB(sk[:nr], 16, 14), engine.SketchDB(sk[nr:], 16, 14)
    got, gf = engine.dist(dbr, dbq, KMERS, TBL)
    want, wf = oracle.query(sk[:nr], sk[nr:], KMERS, 16, 14, TBL, threads=8)
    assert gf == wf and np.abs(got.cpu().numpy() - want).max() <= 1e-6
    a, b = sorted(int(x) for x in rng.integers(0, n - nr + 1, size=2))
    part = engine.dist(dbr, dbq, KMERS, TBL, q_begin=a, q_end=b)[0]
    assert torch.equal(part, got[a * nr:b * nr])
    for d in (db, dbr, dbq):
        d.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_float_patterns_kernel2(seed):
    """Arbitrary float32 bit patterns (denormals, huge, negative, NaN) through assign / edges /
    sweeps: same branches as the un-fused float32 CPU statement."""
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    n = int(rng.integers(3, 300))
    rows = n * (n - 1) // 2
    bits = rng.integers(0, 2 ** 32, size=(rows, 2), dtype=np.uint64).astype(np.uint32)
    d = bits.view(np.float32).copy()
    half = rng.random(rows) < 0.7
    d[half] = (rng.random((int(half.sum()), 2)) * 0.3).astype(np.float32)     # mostly sane values
    d = np.ascontiguousarray(d)
    for slope in (0, 1, 2):
        xm, ym = float(rng.uniform(0.01, 0.3)), float(rng.uniform(0.01, 0.3))
        with np.errstate(all="ignore"):
            assert np.array_equal(poppunk_refine.assignThreshold(d, slope, xm, ym),
                                  oracle.assign_threshold(d, slope, xm, ym))
            assert np.array_equal(poppunk_refine.edgeThreshold_array(d, slope, xm, ym),
                                  oracle.edge_threshold(d, slope, xm, ym))
    finite = np.where(np.isfinite(d), d, np.float32(9.0)).astype(np.float32)
    offs = np.sort(rng.uniform(-0.05, 0.3, size=7))
    gi, gj, go = poppunk_refine.thresholdIterate1D_arrays(finite, offs, 2, 0.05, 0.07, 0.25, 0.3)
    wi, wj, wo = oracle.threshold_iterate_1d(finite, offs, 2, 0.05, 0.07, 0.25, 0.3)
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)


@pytest.mark.parametrize("seed", [3, 4])
def test_random_boundary_sweeps(seed):
    """tools/soak_sweeps.py's cases (thresholdIterate1D / 2D on random matrices, offsets, slopes and directions, rows
    planted on the boundaries, with and without the classify pass's bisection) against the oracle, element for element."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "soak_sweeps", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_sweeps.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(60):
        desc, msgs = mod.case(rng)
        assert not msgs, desc + ": " + "; ".join(msgs)
