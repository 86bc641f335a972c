"""Kernel 1 against the CPU oracle AT BASELINE.json's sizes (the cache-blocked oracle does 80 M
pairs/s on 16 threads, so every row of every config can be compared, not sampled):

  config 2   1 000 genomes self                       every row, counts bit-exact
  config 3  10 000 genomes self                       all 49 995 000 rows + failed-fit count
  config 4  50 000 queries x 10 000 refs              all 5e8 rows, in query bands
  config 5 100 000 genomes self, fused edge list      one band of query rows, edge for edge

Bar: match counts bit-identical; distances within 1e-6 (observed: a handful of rows differ by one
float32 ulp, the rest are identical); failed-fit counts and edge lists identical.
"""
import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import engine, pp_sketchlib, synth

pytestmark = pytest.mark.gpu

KMERS = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
TOL = 1e-6
THREADS = 16


def _device_sketches(n, seed, kmers=KMERS, sketchsize64=16):
    import torch
    t = synth.make_sketches_device(n, kmers, sketchsize64=sketchsize64, seed=seed, device="cuda:0",
                                    chunk=8192 if sketchsize64 <= 16 else 1024)
    sk = t.cpu().numpy().view(np.uint64)
    del t
    torch.cuda.empty_cache()
    return sk


def _compare(got, want, what):
    diff = np.abs(got - want)
    n_diff = int(np.count_nonzero(diff))
    worst = float(diff.max()) if diff.size else 0.0
    print("%s: %d rows, max |d - oracle| = %.3g, %d values differ" % (what, len(got), worst, n_diff))
    assert worst <= TOL, what
    # "a few rows in a million differ by one float32 ulp", not "agree to 1e-6 everywhere by luck"
    assert n_diff <= max(20, len(got) // 20000), (what, n_diff)
    return worst, n_diff


def test_config2_1000_genomes_self_exact():
    sk, _ = synth.make_sketches(1000, KMERS)
    tbl = synth.random_match_table(KMERS)
    counts, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, counts=True)
    assert counts.shape == (499500, 5)
    assert np.array_equal(counts, oracle.match_counts(sk, None, 16, 14, threads=THREADS))
    got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, tbl, threads=THREADS)
    assert gf == wf == 0
    _compare(got, want, "config 2 (1 000 self)")
    jac, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl, jaccard=True)
    wj, _ = oracle.query(sk, None, KMERS, 16, 14, tbl, jaccard=True, threads=THREADS)
    assert np.array_equal(jac, wj)


@pytest.mark.parametrize("related", [True, False])
def test_config3_10000_genomes_self_every_row(related):
    """All 49 995 000 rows of the bench workload (related=True: bench.py's own sketches) and of
    the unrelated-cluster variant (49 750 000 failing fits: the failed-fit path at full size)."""
    sk, _ = synth.make_sketches(10000, KMERS, related=related)
    tbl = synth.random_match_table(KMERS)
    got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, tbl, threads=THREADS)
    assert got.shape == (49995000, 2)
    assert gf == wf
    if related:
        assert wf == 0
        _compare(got, want, "config 3 (10 000 self)")
    else:
        assert wf > 49000000
        assert np.array_equal(got, want)


def test_config4_50000_queries_x_10000_refs_in_bands():
    """ref sketches resident, 50 000 queries, every one of the 5e8 rows (row = q*n_ref + r)."""
    import torch
    allsk = _device_sketches(60000, seed=4)
    ref, qry = allsk[:10000], allsk[10000:]
    tbl = synth.random_match_table(KMERS)
    dr = engine.SketchDB(ref, 16, 14)
    dq = engine.SketchDB(qry, 16, 14)
    worst, n_diff, failed_gpu, failed_cpu = 0.0, 0, 0, 0
    band = 6400
    for qb in range(0, 50000, band):
        qe = min(50000, qb + band)
        d, f = engine.dist(dr, dq, KMERS, tbl, q_begin=qb, q_end=qe)
        got = d.cpu().numpy()
        failed_gpu += int(f.item())
        del d
        want, wf = oracle.query(ref, qry[qb:qe], KMERS, 16, 14, tbl, threads=THREADS)
        failed_cpu += wf
        diff = np.abs(got - want)
        worst = max(worst, float(diff.max()))
        n_diff += int(np.count_nonzero(diff))
    dr.close()
    dq.close()
    torch.cuda.empty_cache()
    print("config 4 (50 000 x 10 000): 500000000 rows, max |d - oracle| = %.3g, %d values differ"
          % (worst, n_diff))
    assert failed_gpu == failed_cpu
    assert worst <= TOL and n_diff <= 25000


def test_config5_100000_genomes_one_band_fused_edges():
    """100 000 genomes self: the fused distance -> boundary -> edge-list kernel on one band of
    query rows (what one of 8 GPUs does, engine.edges_sharded) against oracle.query +
    assign_threshold on exactly those rows."""
    import torch
    n = 100000
    sk = _device_sketches(n, seed=5)
    tbl = synth.random_match_table(KMERS)
    db = engine.SketchDB(sk, 16, 14)
    # boundary through the 2 % quantiles of a subsample's distances
    sub, _ = oracle.query(sk[:1500], None, KMERS, 16, 14, tbl, threads=THREADS)
    x_max, y_max = synth.boundary_for_quantile(sub, 0.02)
    total_edges = 0
    for qb, qe in ((0, 256), (49984, 50624), (99712, n)):        # first rows, a mid band, the last rows
        # the band's rows: query q in [qb, qe) against every ref r > q
        rect, _ = oracle.query(sk, sk[qb:qe], KMERS, 16, 14, tbl, threads=THREADS)   # row = (q-qb)*n + r
        a = oracle.assign_threshold(rect, 2, x_max, y_max, threads=THREADS).reshape(qe - qb, n)
        del rect
        for inclusive in (True, False):
            e, nf = engine.dist_edges(db, None, KMERS, tbl, slope=2, x_max=x_max, y_max=y_max,
                                      inclusive=inclusive, q_begin=qb, q_end=qe, cap=1 << 20)
            got = e.cpu().numpy()
            qq, rr = np.nonzero((a <= 0) if inclusive else (a < 0))
            sel = rr > qq + qb
            want = np.stack([qq[sel] + qb, rr[sel]], axis=1).astype(np.int64)
            assert int(nf.item()) == 0
            assert np.array_equal(got, want), (qb, qe, inclusive, len(got), len(want))
            total_edges += len(want)
    assert total_edges > 1000
    db.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nk", [5, 6])
def test_default_sketch_size_many_ref_tiles(nk, ppk_option):
    """PopPUNK's DEFAULT sketch size (s = 9 984: sketchsize64 156, 14-bit counts -> the three-dword count
    register) on a job with 11 ref tiles and a ragged right edge: 3.5 M pairs of 780 / 936 blocks each.
    Counts bit-identical, distances within 1e-6 on every row, query bands equal to the whole job, the
    fused edge list equal to the oracle's, ref x query as well."""
    import torch
    n = 2650
    kmers = np.asarray([13, 17, 21, 25, 29] if nk == 5 else [13, 16, 19, 22, 25, 28], dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk = _device_sketches(n, 600 + nk, kmers, 156)          # (drawn on the GPU: 20 s with numpy on the host)
    counts, _ = pp_sketchlib.query_arrays(sk, None, kmers, 156, 14, counts=True)
    assert np.array_equal(counts, oracle.match_counts(sk, None, 156, 14, threads=THREADS))
    del counts
    want, wf = oracle.query(sk, None, kmers, 156, 14, tbl, threads=THREADS)
    db = engine.SketchDB(sk, 156, 14)
    # long sketches take the k-split path at any size by default (round 5); the tile kernel itself with "ksplit" 0:
    # the same bits either way
    by_units, gf_u = engine.dist(db, None, kmers, tbl)
    assert engine._lib.lib().ppk_last_kernel_name().decode().endswith("k-split fused,fit from parts>")
    ppk_option("ksplit", 0)
    whole, gf = engine.dist(db, None, kmers, tbl)
    assert engine._lib.lib().ppk_last_kernel_name().decode().endswith("lds-dma>")
    assert int(gf.item()) == wf == int(gf_u.item()) and torch.equal(by_units.view(torch.int32), whole.view(torch.int32))
    del by_units
    _compare(whole.cpu().numpy(), want, "s=9984 nk=%d, %d genomes self" % (nk, n))
    cuts = [0, 37, 1024, 1111, 2600, n]
    pieces = [engine.dist(db, None, kmers, tbl, q_begin=a, q_end=b)[0] for a, b in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(torch.cat(pieces), whole)
    x_max, y_max = synth.boundary_for_quantile(want, 0.05)
    e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)      # (the tile kernel: "ksplit" is 0)
    assert np.array_equal(e.cpu().numpy(), oracle.edge_threshold(want, 2, x_max, y_max))
    # the fused edge list through the k-split path (long sketches: the default), whole and in bands
    ppk_option("ksplit", 1200)
    e2, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
    assert torch.equal(e, e2)
    parts = [engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, inclusive=False, q_begin=a, q_end=b)[0]
             for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(torch.cat(parts).cpu().numpy(), oracle.edge_threshold(want, 2, x_max, y_max, inclusive=False))
    db.close()
    nr = 1500
    got, gf2 = pp_sketchlib.query_arrays(sk[:nr], sk[nr:], kmers, 156, 14, tbl)
    want2, wf2 = oracle.query(sk[:nr], sk[nr:], kmers, 156, 14, tbl, threads=THREADS)
    assert gf2 == wf2
    _compare(got, want2, "s=9984 nk=%d, %d x %d" % (nk, nr, n - nr))


def test_400000_genomes_one_device_call_spans_several_dispatches():
    """400 000 genomes against themselves are 9.8 M pair tiles, more than one dispatch holds (2^32 work-items =
    8.4 M tiles of 512 threads): ppk_launch_dist sends the band out as several launches.  The fused edge list of
    ONE device call must equal the host call's, which works through the band in pieces of its own (another
    cut), rows ascending, and sampled pairs -- edges and non-edges -- must agree with the oracle."""
    import torch
    n = 400000
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk_t = synth.make_sketches_device(n, kmers, device="cuda:0")
    db = engine.SketchDB(sk_t, 16, 14, device=0)
    sub = engine.SketchDB(synth.make_sketches_device(2000, kmers, device="cuda:0"), 16, 14, device=0)
    d_sub, _ = engine.dist(sub, None, kmers, tbl)
    x_max, y_max = synth.boundary_for_quantile(d_sub.cpu().numpy(), 0.02)
    sub.close()
    e1, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=32 << 20)
    e1 = e1.cpu().numpy()
    e2, nf = engine.edges_host(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, cap=32 << 20)
    assert nf == 0 and len(e1) > 1000000 and np.array_equal(e1, e2)
    key = e1[:, 0] * n + e1[:, 1]
    assert bool(np.all(e1[:, 0] < e1[:, 1])) and bool(np.all(np.diff(key) > 0))
    rng = np.random.Generator(np.random.PCG64(5))
    n_clusters = n // 50
    sample = [tuple(int(v) for v in e1[i]) for i in rng.choice(len(e1), 60, replace=False)]
    for _ in range(60):          # pairs of one cluster (members c, c + n_clusters, ...): edges and non-edges
        c = int(rng.integers(0, n_clusters))
        sample.append(tuple(sorted(int(v) for v in rng.choice(np.arange(c, n, n_clusters), 2, replace=False))))
    sample += [tuple(sorted(int(v) for v in rng.choice(n, 2, replace=False))) for _ in range(30)]
    # pairs late in the triangle: rows that only the later launches cover
    sample += [tuple(sorted((int(a), int(b)))) for a, b in zip(rng.integers(390000, n, 30), rng.integers(380000, 390000, 30))]
    for a, b in sample:
        pair = sk_t[[a, b]].cpu().numpy().view(np.uint64)
        d, _ = oracle.query(pair[:1], pair[1:], kmers, 16, 14, tbl, threads=1)
        want = bool(oracle.edge_threshold(d, 2, x_max, y_max, n_ref=1, inclusive=True).shape[0])
        pos = np.searchsorted(key, a * n + b)
        assert bool(pos < len(key) and key[pos] == a * n + b) == want, (a, b)
    db.close()
    del sk_t
    torch.cuda.empty_cache()
    from poppunk_amd import _lib
    _lib.lib().ppk_release_scratch()


@pytest.mark.parametrize("knn", [10, 32])
def test_neighbours_of_40000_genomes_staged_flow_against_brute_force_rows(knn):
    """Jobs of 16 384 rows or more run the neighbour mode staged (a short opening, a cut, growing pieces, cuts at
    4 n k): 40 000 genomes with the real thresholds, sampled samples against a brute-force row of the oracle
    (all 40 000 distances of that sample, stable order, ties to the lower index)."""
    n = 40000
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk_t = synth.make_sketches_device(n, kmers, device="cuda:0")
    db = engine.SketchDB(sk_t, 16, 14, device=0)
    info = {}
    oi, oj, od = engine.knn_from_sketches(db, kmers, tbl, knn, dist_col=0, method="tiles", info=info)
    oj, od = oj.cpu().numpy().reshape(n, knn), od.cpu().numpy().reshape(n, knn)
    assert np.array_equal(oi.cpu().numpy(), np.repeat(np.arange(n), knn))
    assert info["candidates"] < 12 * n * knn            # the list was cut on the way (un-staged: hundreds per sample and k)
    host = sk_t.cpu().numpy().view(np.uint64)
    rng = np.random.Generator(np.random.PCG64(knn))
    for r in rng.choice(n, size=10, replace=False).tolist() + [0, n - 1]:
        d, _ = oracle.query(host, host[r:r + 1], kmers, 16, 14, tbl, threads=8)
        col = d[:, 0]
        order = np.argsort(col, kind="stable")
        order = order[order != r][:knn]
        assert np.array_equal(oj[r], order) and np.array_equal(od[r], col[order]), r
    db.close()


def _mixture_matrix(n, seed):
    """A distance matrix of n genomes' worth of rows with PopPUNK's shape: a tight within-strain cloud near the origin,
    a broad between-strain cloud, a few exact repeats (ties) and exact zeros; float32, non-negative."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = n * (n - 1) // 2
    d = np.empty((rows, 2), dtype=np.float32)
    step = 1 << 22
    for lo in range(0, rows, step):
        m = min(step, rows - lo)
        near = rng.random(m) < 0.15
        x = np.where(near, rng.random(m) * 0.004, 0.01 + rng.random(m) * 0.03)
        y = np.where(near, rng.random(m) * 0.05, 0.1 + rng.random(m) * 0.4)
        d[lo:lo + m, 0] = x
        d[lo:lo + m, 1] = y
    d[rng.integers(0, rows, 5000)] = 0.0
    d[rng.integers(0, rows, 5000)] = d[rng.integers(0, rows, 1)]
    return d


@pytest.mark.parametrize("n", [10000, 12000])
def test_boundary_sweeps_at_baseline_size_element_for_element(n):
    """SURVEY 8(f1) at the size refine runs it: thresholdIterate1D (40 offsets) and thresholdIterate2D (20 offsets) on
    the long matrix of 10 000 genomes (49 995 000 rows: 32-bit (row, first boundary) values, the filtered classify
    pass, the radix sort of ~17 % of the rows) and of 12 000 genomes (71 994 000 rows: row and boundary index no longer
    fit 32 bits, the 64-bit value path) against the oracle's restatement of src/boundary.cpp:154-237."""
    import torch
    d = _mixture_matrix(n, seed=n)
    dt = torch.from_numpy(d).cuda()
    m0, m1 = np.asarray([0.002, 0.02], dtype=np.float32), np.asarray([0.012, 0.14], dtype=np.float32)
    offs = np.linspace(0.0, float(np.linalg.norm(m1 - m0)), 40)
    gi, gj, go = engine.threshold_iterate_1d_dev(dt, offs, 2, m0[0], m0[1], m1[0], m1[1])
    wi, wj, wo = oracle.threshold_iterate_1d(d, offs, 2, m0[0], m0[1], m1[0], m1[1])
    assert len(wi) > len(d) // 10
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gj.cpu().numpy(), wj) and np.array_equal(go.cpu().numpy(), wo)
    del gi, gj, go
    xm = np.linspace(0.003, 0.02, 20).astype(np.float32)
    gi, gj, go = engine.threshold_iterate_2d_dev(dt, xm, 0.2)
    wi, wj, wo = oracle.threshold_iterate_2d(d, xm, 0.2)
    assert len(wi) > len(d) // 10
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gj.cpu().numpy(), wj) and np.array_equal(go.cpu().numpy(), wo)
    del dt, gi, gj, go
    torch.cuda.empty_cache()
