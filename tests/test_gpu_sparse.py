"""poppunk_refine.extend / lowerRank on the MI355X against the oracle's restatement of src/extend.cpp:52-246
(parity unpinned: extend.cpp cannot be built here).  Distances are drawn from a few values so that ties --
whose order is the whole content of these functions -- are everywhere."""
import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import poppunk_refine

pytestmark = pytest.mark.gpu


def _square(rng, n, levels):
    v = rng.integers(1, levels + 1, size=(n, n)).astype(np.float32) / np.float32(levels * 4)
    v = np.triu(v, 1)
    return v + v.T


def _knn_coo(sq, k):
    i, j, d = poppunk_refine.get_kNN_distances(sq, k)
    return np.asarray(i, dtype=np.int64), np.asarray(j, dtype=np.int64), np.asarray(d, dtype=np.float32)


def _same(got, want):
    gi, gj, gd = got
    wi, wj, wd = want
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj)
    assert gd.dtype == np.float32 and np.array_equal(gd, wd)


@pytest.mark.parametrize("n,depth,levels", [(40, 6, 5), (257, 10, 50), (1000, 12, 7), (3, 2, 2)])
@pytest.mark.parametrize("unique", [False, True])
@pytest.mark.parametrize("recip", [False, True])
def test_lower_rank(n, depth, levels, unique, recip):
    rng = np.random.Generator(np.random.PCG64(n * 7 + depth))
    sq = _square(rng, n, levels)
    coo = _knn_coo(sq, min(depth, n - 1))
    for knn in (1, 2, 3, depth + 2):
        for eps in ((0.0, 1e-3, 0.06) if unique else (0.0,)):
            want = oracle.lower_rank(*coo, n, knn, reciprocal_only=recip, count_unique_distances=unique, epsilon=eps)
            got = poppunk_refine.lowerRank_arrays(coo, n, knn, recip, unique, eps)
            _same(got, want)
    i, j, d = poppunk_refine.lowerRank(coo, n, 1, recip, unique, 0.001, 4)
    assert isinstance(i, list) and isinstance(d, list) and len(i) == len(j) == len(d)


def test_lower_rank_rows_with_self_entries_gaps_and_bad_input():
    # rows 1 and 4 are empty, row 2 holds its own sample, row 3 has equal distances in input order
    ri = np.asarray([0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 5], dtype=np.int64)
    rj = np.asarray([3, 1, 2, 2, 0, 5, 0, 1, 2, 5, 3], dtype=np.int64)
    rd = np.asarray([.3, .1, .2, 0., .2, .1, .5, .5, .5, .5, .5], dtype=np.float32)
    for knn in (0, 1, 2, 5):
        for unique in (False, True):
            for recip in (False, True):
                want = oracle.lower_rank(ri, rj, rd, 6, knn, recip, unique, 1e-5)
                _same(poppunk_refine.lowerRank_arrays((ri, rj, rd), 6, knn, recip, unique, 1e-5), want)
    # lists, as pybind accepts them; an empty matrix
    got = poppunk_refine.lowerRank((ri.tolist(), rj.tolist(), rd.tolist()), 6, 1)
    want = oracle.lower_rank(ri, rj, rd, 6, 1)
    assert got[0] == want[0].tolist() and got[1] == want[1].tolist() and np.allclose(got[2], want[2])
    assert poppunk_refine.lowerRank(([], [], []), 4, 2) == ([], [], [])
    with pytest.raises(RuntimeError, match="ascending"):
        poppunk_refine.lowerRank_arrays((ri[::-1].copy(), rj, rd), 6, 1)
    with pytest.raises(RuntimeError, match="ascending"):
        poppunk_refine.lowerRank_arrays((ri + 3, rj, rd), 6, 1)
    with pytest.raises(TypeError):
        poppunk_refine.lowerRank_arrays(ri, 6, 1)


@pytest.mark.parametrize("n_ref,n_qry,depth,levels", [(30, 7, 4, 4), (500, 61, 9, 6), (64, 200, 5, 40), (5, 1, 3, 2),
                                                      (1, 3, 2, 3)])
def test_extend(n_ref, n_qry, depth, levels):
    rng = np.random.Generator(np.random.PCG64(n_ref * 3 + n_qry))
    rr = _square(rng, n_ref, levels)
    coo = _knn_coo(rr, min(depth, n_ref - 1)) if n_ref > 1 else (np.zeros(0, np.int64), np.zeros(0, np.int64),
                                                                  np.zeros(0, np.float32))
    qq = _square(rng, n_qry, levels)
    qr = (rng.integers(1, levels + 1, size=(n_ref, n_qry)).astype(np.float32) / np.float32(levels * 4))
    for knn in (1, depth, depth + 3):
        want = oracle.extend(*coo, qq, qr, knn)
        _same(poppunk_refine.extend_arrays(coo, qq, qr, knn), want)
    # the caller's shape (PopPUNK/models.py:1355-1372): qr arrives as a transposed view, the sparse matrix as
    # scipy-style attributes; the result feeds lowerRank
    qr_view = np.ascontiguousarray(qr.T).T
    assert not qr_view.flags["C_CONTIGUOUS"] or min(qr.shape) == 1
    i, j, d = poppunk_refine.extend((coo[0], coo[1], coo[2]), qq, qr_view, depth, 2)
    want = oracle.extend(*coo, qq, qr, depth)
    assert i == want[0].tolist() and j == want[1].tolist() and np.array_equal(np.asarray(d, dtype=np.float32), want[2])
    low = poppunk_refine.lowerRank_arrays((i, j, d), n_ref + n_qry, 2)
    _same(low, oracle.lower_rank(want[0], want[1], want[2], n_ref + n_qry, 2))
    with pytest.raises(TypeError):
        poppunk_refine.extend(coo, qq.astype(np.float64), qr, 2)
    with pytest.raises(RuntimeError):
        poppunk_refine.extend(coo, qq[:-1], qr, 2) if n_qry > 1 else poppunk_refine.extend(coo, np.zeros((2, 2), np.float32), qr, 2)


def test_extend_then_ranks_of_a_lineage_model_at_size():
    """20 000 references with 10 neighbours each + 300 queries (6.5 M candidate distances): spot rows
    against the rule written out ((distance, query side first, place) order, self skipped), and the invariants
    of the whole result."""
    rng = np.random.Generator(np.random.PCG64(99))
    n_ref, n_qry, depth = 20000, 300, 10
    ri = np.repeat(np.arange(n_ref, dtype=np.int64), depth)
    rj = (ri + rng.integers(1, n_ref, size=ri.size)) % n_ref
    rd = np.sort(rng.integers(1, 2000, size=(n_ref, depth)).astype(np.float32) / np.float32(8000), axis=1).ravel()
    qq = _square(rng, n_qry, 500)
    qr = rng.integers(1, 2000, size=(n_ref, n_qry)).astype(np.float32) / np.float32(8000)
    i, j, d = poppunk_refine.extend_arrays((ri, rj, rd), qq, qr, depth)
    assert len(i) == depth * (n_ref + n_qry) and np.array_equal(i, np.repeat(np.arange(n_ref + n_qry), depth))
    assert not np.any(i == j)
    dd = d.reshape(-1, depth)
    assert bool(np.all(np.diff(dd, axis=1) >= 0))
    rows = np.concatenate([rng.integers(0, n_ref, 40), n_ref + rng.integers(0, n_qry, 40)])
    for r in rows.tolist():
        if r < n_ref:
            sel = slice(r * depth, (r + 1) * depth)
            cand = ([(qr[r, c], 0, c, n_ref + c) for c in range(n_qry)]
                    + [(rd[sel][p], 1, p, int(rj[sel][p])) for p in range(depth)])
            cand = [c for c in sorted(cand) if c[3] != r][:depth]
            wj, wd = np.asarray([c[3] for c in cand]), np.asarray([c[0] for c in cand], dtype=np.float32)
        else:
            q = r - n_ref
            order_q = np.argsort(qq[q], kind="stable")
            order_r = np.argsort(qr[:, q], kind="stable")
            cand = [(qq[q][c], 0, c, n_ref + c) for c in order_q[:depth + 1]] + [(qr[c, q], 1, c, c) for c in order_r[:depth + 1]]
            cand = [c for c in sorted(cand) if c[3] != r][:depth]
            wj, wd = np.asarray([c[3] for c in cand]), np.asarray([c[0] for c in cand], dtype=np.float32)
        assert np.array_equal(j[r * depth:(r + 1) * depth], wj), r
        assert np.array_equal(d[r * depth:(r + 1) * depth], wd), r


def test_sparse_fuzz():
    """300 random small problems: empty sides, kNN beyond what a row holds, rows without entries, sparse
    rows that name their own sample, few distinct distances."""
    rng = np.random.Generator(np.random.PCG64(4242))
    for case in range(300):
        n_ref = int(rng.integers(0, 25))
        n_qry = int(rng.integers(0, 9))
        levels = int(rng.integers(1, 6))
        nnz_per = rng.integers(0, 7, size=n_ref)
        ri = np.repeat(np.arange(n_ref, dtype=np.int64), nnz_per)
        rj = rng.integers(0, max(n_ref, 1), size=ri.size).astype(np.int64)
        rd = (rng.integers(0, levels + 1, size=ri.size) / np.float32(8)).astype(np.float32)
        qq = _square(rng, n_qry, levels) if n_qry else np.zeros((0, 0), np.float32)
        qr = (rng.integers(0, levels + 1, size=(n_ref, n_qry)) / np.float32(8)).astype(np.float32)
        knn = int(rng.integers(0, 9))
        got = poppunk_refine.extend_arrays((ri, rj, rd), qq, qr, knn)
        want = oracle.extend(ri, rj, rd, qq, qr, knn)
        _same(got, want)
        n = n_ref + n_qry
        unique, recip = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        eps = float(rng.choice([0.0, 0.125, 0.2, 1e-6]))
        k2 = int(rng.integers(0, 6))
        _same(poppunk_refine.lowerRank_arrays(want, n, k2, recip, unique, eps),
              oracle.lower_rank(*want, n, k2, recip, unique, eps))
        _same(poppunk_refine.lowerRank_arrays((ri, rj, rd), n_ref, k2, recip, unique, eps),
              oracle.lower_rank(ri, rj, rd, n_ref, k2, recip, unique, eps))


def test_lineage_ranks_from_distances_from_sketches_and_extended(tmp_path):
    """poppunk_amd.models.LineageRanks: the matrices of LineageFit.fit (PopPUNK/models.py:1188-1237) from a
    distance matrix, the same straight from the sketches, then queries added (models.py:1334-1385); each
    compared with the composition of the oracle's pieces."""
    from poppunk_amd import models, pp_sketchlib, sketchdb, synth
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(420, kmers, cluster_size=30, seed=3)
    tbl = synth.random_match_table(kmers)
    names = ["n%03d" % i for i in range(420)]
    db = str(tmp_path / "db")
    sketchdb.save_npz(db, names, kmers, sk, 16, 14, random_table=tbl)
    pp_sketchlib.clear_cache()
    n_ref = 300
    X = pp_sketchlib.queryDatabase(db, db, names[:n_ref], names[:n_ref], kmers.tolist(), True, False, 1, True, 0)
    for recip, unique in ((False, False), (True, True)):
        a = models.LineageRanks([1, 2, 3], 4, recip, unique, 1e-4, dist_col=1)
        y = a.fit(X)
        b = models.LineageRanks([1, 2, 3], 4, recip, unique, 1e-4, dist_col=1)
        assert b.fit_from_database(db, names[:n_ref], kmers.tolist()) == y
        depth = a.max_search_depth
        assert depth == 8
        oi, oj, od = oracle.knn(oracle.long_to_square(X[:, 1]), depth)
        for m in (a, b):
            assert np.array_equal(m.nn_dists.row, oi) and np.array_equal(m.nn_dists.col, oj)
            assert np.array_equal(m.nn_dists.data, np.maximum(od, np.float32(1e-10)))
            for rank in (1, 2, 3):
                wi, wj, wd = oracle.lower_rank(oi, oj, od, n_ref, rank, recip, unique, 1e-4)
                lr = m.lower_rank_dists[rank]
                assert np.array_equal(lr.row, wi) and np.array_equal(lr.col, wj)
                assert np.array_equal(lr.data, np.maximum(wd, np.float32(1e-10)))
            assert m.assign(1) == list(zip(m.lower_rank_dists[1].row.tolist(), m.lower_rank_dists[1].col.tolist()))
        # queries join the fitted model
        qq = pp_sketchlib.queryDatabase(db, db, names[n_ref:], names[n_ref:], kmers.tolist(), True, False, 1, True, 0)
        qr = pp_sketchlib.queryDatabase(db, db, names[:n_ref], names[n_ref:], kmers.tolist(), True, False, 1, True, 0)
        a.extend(qq, qr)
        floor = np.float32(1e-10)
        qq_sq = np.maximum(oracle.long_to_square(qq[:, 1]), floor)
        qr_rect = np.maximum(qr[:, 1].reshape(120, n_ref).T, floor)
        ei, ej, ed = oracle.extend(oi, oj, np.maximum(od, floor), qq_sq, qr_rect, depth)
        assert np.array_equal(a.nn_dists.row, ei) and np.array_equal(a.nn_dists.col, ej)
        assert np.array_equal(a.nn_dists.data, np.maximum(ed, floor))
        wi, wj, wd = oracle.lower_rank(ei, ej, ed, 420, 2, recip, unique, 1e-4)
        assert np.array_equal(a.lower_rank_dists[2].row, wi) and np.array_equal(a.lower_rank_dists[2].data, np.maximum(wd, floor))
    pp_sketchlib.clear_cache()


@pytest.mark.parametrize("n_ref,n_qry,related", [(600, 150, True), (300, 500, True), (520, 90, False), (40, 3, True)])
def test_extend_from_sketches_equals_extend_on_the_dense_matrices(n_ref, n_qry, related):
    """ppk_extend_sketches: the tiles deliver every reference's nearest queries and every query's nearest
    references and queries; merged with the sparse rows exactly as extend merges the dense rectangle and
    square (which are computed here only to state the expected result).  Unrelated data: ties everywhere."""
    from poppunk_amd import engine, pp_sketchlib, synth
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(n_ref + n_qry, kmers, cluster_size=25, seed=n_ref + n_qry, related=related)
    tbl = synth.random_match_table(kmers)
    ref, qry = sk[:n_ref], sk[n_ref:]
    rr, _ = pp_sketchlib.query_arrays(ref, None, kmers, 16, 14, tbl)
    qq, _ = pp_sketchlib.query_arrays(qry, None, kmers, 16, 14, tbl)
    qr, _ = pp_sketchlib.query_arrays(ref, qry, kmers, 16, 14, tbl)
    rdb, qdb = engine.SketchDB(ref, 16, 14), engine.SketchDB(qry, 16, 14)
    for col, depth in ((0, 6), (1, 3)):
        coo = oracle.knn(oracle.long_to_square(rr[:, col]), min(depth, n_ref - 1))
        qq_sq = oracle.long_to_square(qq[:, col]) if n_qry > 1 else np.zeros((n_qry, n_qry), np.float32)
        qr_rect = np.ascontiguousarray(qr[:, col].reshape(n_qry, n_ref).T)
        for knn in (1, depth, depth + 4):
            want = oracle.extend(*coo, qq_sq, qr_rect, knn)
            got = engine.extend_from_sketches(coo, rdb, qdb, kmers, tbl, knn, dist_col=col)
            _same(got, want)
            _same(engine.extend_from_sketches(coo, [rdb] * 3, [qdb] * 3, kmers, tbl, knn, dist_col=col), want)
            _same(poppunk_refine.extend_arrays(coo, qq_sq, qr_rect, knn), want)
    rdb.close()
    qdb.close()


def test_lineage_ranks_extended_from_databases(tmp_path, monkeypatch):
    from poppunk_amd import models, pp_sketchlib, sketchdb, synth
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(500, kmers, cluster_size=25, seed=8)
    tbl = synth.random_match_table(kmers)
    names = ["m%03d" % i for i in range(500)]
    rdb, qdb = str(tmp_path / "refs"), str(tmp_path / "queries")
    sketchdb.save_npz(rdb, names[:380], kmers, sk[:380], 16, 14, random_table=tbl)
    sketchdb.save_npz(qdb, names[380:], kmers, sk[380:], 16, 14, random_table=tbl)
    pp_sketchlib.clear_cache()
    klist = kmers.tolist()
    a = models.LineageRanks([1, 2], 3, dist_col=0)
    b = models.LineageRanks([1, 2], 3, dist_col=0)
    assert a.fit_from_database(rdb, names[:380], klist) == b.fit_from_database(rdb, names[:380], klist)
    qq = pp_sketchlib.queryDatabase(qdb, qdb, names[380:], names[380:], klist, True, False, 1, True, 0)
    qr = pp_sketchlib.queryDatabase(rdb, qdb, names[:380], names[380:], klist, True, False, 1, True, 0)
    ya = a.extend(qq, qr)
    monkeypatch.setenv("PPK_DEVICES", "0,0")
    yb = b.extend_from_databases(rdb, qdb, names[:380], names[380:], klist)
    monkeypatch.delenv("PPK_DEVICES")
    assert ya == yb and len(ya) > 500
    for m, w in ((b.nn_dists, a.nn_dists), (b.lower_rank_dists[2], a.lower_rank_dists[2])):
        assert np.array_equal(m.row, w.row) and np.array_equal(m.col, w.col) and np.array_equal(m.data, w.data)
    pp_sketchlib.clear_cache()
