"""SURVEY.md 8f rank 2 on a real MI355X: long <-> square transforms and kNN vs scipy/numpy."""
import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import engine, poppunk_refine, pp_sketchlib, qc, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2, 3, 63, 64, 65, 200, 1000])
def test_long_to_square_and_back(n):
    rng = np.random.Generator(np.random.PCG64(n))
    v = rng.random(n * (n - 1) // 2).astype(np.float32)
    sq = pp_sketchlib.longToSquare(v.reshape(-1, 1), 2)           # PopPUNK passes distMat[:, [0]]
    assert sq.shape == (n, n) and sq.dtype == np.float32
    assert np.array_equal(sq, oracle.long_to_square(v))
    assert np.array_equal(np.diag(sq), np.zeros(n)) and np.array_equal(sq, sq.T)
    back = pp_sketchlib.squareToLong(sq, 2)
    assert np.array_equal(back, v) and np.array_equal(back, oracle.square_to_long(sq))


@pytest.mark.parametrize("n_ref,n_qry", [(2, 2), (5, 3), (64, 65), (130, 70), (3, 200)])
def test_long_to_square_multi(n_ref, n_qry):
    rng = np.random.Generator(np.random.PCG64(n_ref * 1000 + n_qry))
    rr = rng.random(n_ref * (n_ref - 1) // 2).astype(np.float32)
    qr = rng.random(n_ref * n_qry).astype(np.float32)
    qq = rng.random(n_qry * (n_qry - 1) // 2).astype(np.float32)
    got = pp_sketchlib.longToSquareMulti(rr.reshape(-1, 1), qr.reshape(-1, 1), qq.reshape(-1, 1), 1)
    assert np.array_equal(got, oracle.long_to_square_multi(rr, qr, qq))
    assert np.array_equal(got, got.T)


def test_transform_errors():
    with pytest.raises(TypeError):
        pp_sketchlib.longToSquare(np.zeros(3), 1)                      # float64
    with pytest.raises(RuntimeError):
        pp_sketchlib.longToSquare(np.zeros(4, dtype=np.float32), 1)    # 4 != n(n-1)/2
    with pytest.raises(RuntimeError):
        pp_sketchlib.squareToLong(np.zeros((2, 3), dtype=np.float32), 1)


@pytest.mark.parametrize("lane_lists", [0, 1])
@pytest.mark.parametrize("n,k", [(5, 2), (64, 3), (300, 10), (300, 1), (4, 7), (1000, 20), (70, 32),
                                 (70, 33), (200, 40), (130, 8), (130, 9), (700, 64), (66, 65), (257, 5)])
def test_knn(n, k, lane_lists, ppk_option):
    """get_kNN_distances (src/extend.cpp:248-289): the one-list-per-wavefront selection (the default up to 64
    neighbours), the per-lane lists before it (option knn_lane_lists) and the segmented sort beyond."""
    ppk_option("knn_lane_lists", lane_lists)
    rng = np.random.Generator(np.random.PCG64(n + k))
    v = (rng.integers(0, 50, size=n * (n - 1) // 2) / 50.0).astype(np.float32)   # many ties
    sq = oracle.long_to_square(v)
    gi, gj, gd = poppunk_refine.get_kNN_distances(sq, k, 0, 4)
    wi, wj, wd = oracle.knn(sq, k)
    assert gi == wi.tolist() and gj == wj.tolist() and gd == wd.tolist()
    assert len(gi) == n * k
    with pytest.raises(TypeError):
        poppunk_refine.get_kNN_distances(sq.astype(np.float64), k)


def test_knn_straight_from_sketches():
    """engine.knn_from_sketches == get_kNN_distances(longToSquare(dist[:, col])) without the square
    matrix (band by band; several bands forced by a small band_items)."""
    from poppunk_amd import engine, synth
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(500, kmers, cluster_size=25, seed=4)
    tbl = synth.random_match_table(kmers)
    dist, _ = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl)
    db = engine.SketchDB(sk, 16, 14)
    for col, k in ((0, 5), (1, 3)):
        sq = oracle.long_to_square(dist[:, col])
        wi, wj, wd = oracle.knn(sq, k)
        for method, band_items in (("bands", 500 * 128), ("square", 1 << 31), ("tiles", 0)):
            gi, gj, gd = engine.knn_from_sketches(db, kmers, tbl, k, dist_col=col, band_items=band_items,
                                                  method=method)
            assert np.array_equal(gi.cpu().numpy(), wi), method
            assert np.array_equal(gd.cpu().numpy(), wd), method
            # ties between equal distances resolve by column index in both
            assert np.array_equal(gj.cpu().numpy(), wj), method
    db.close()


@pytest.mark.parametrize("n,related", [(3000, True), (2300, False), (257, True), (40, True)])
def test_knn_from_tiles_equals_oracle(n, related):
    """Neighbours straight from kernel 1's tiles (MODE_KNN: candidates under tightening per-sample
    bounds, then sort + select) == get_kNN_distances(longToSquare(.)) of the oracle's distances, at a
    size with a dozen ref tiles, diagonal half tiles and a ragged edge; unrelated clusters give
    masses of failed fits (distance 0.0: ties everywhere, resolved by column index); tiny n leaves
    slots unfilled like the reference."""
    from poppunk_amd import engine, synth
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(n, kmers, cluster_size=37, seed=n, related=related)
    tbl = synth.random_match_table(kmers)
    want, _ = oracle.query(sk, None, kmers, 16, 14, tbl, threads=16)
    db = engine.SketchDB(sk, 16, 14)
    for col, k in ((0, 5), (1, 1), (0, 32)):
        sq = oracle.long_to_square(want[:, col])
        wi, wj, wd = oracle.knn(sq, k)
        info = {}
        gi, gj, gd = engine.knn_from_sketches(db, kmers, tbl, k, dist_col=col, method="tiles", info=info)
        assert np.array_equal(gi.cpu().numpy(), wi)
        gd, gj = gd.cpu().numpy(), gj.cpu().numpy()
        # distances within the regression tolerance; where they are bit-equal (all but a handful of
        # rows) the neighbour order -- ties by column index -- is the reference's
        assert np.abs(gd - wd).max() <= 1e-6
        same = gd == wd
        assert same.mean() > 0.999 and np.array_equal(gj[same], wj[same])
        assert 0 < info["candidates"] <= n * (n - 1)
    db.close()


def test_knn_candidates_of_bands_select_to_the_whole_answer():
    """The N-GPU decomposition on one GPU: the candidates of four bands of query rows (any order),
    concatenated and selected, are the neighbours of the whole job."""
    import torch
    from poppunk_amd import engine, synth
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(2600, kmers, cluster_size=40, seed=26)
    tbl = synth.random_match_table(kmers)
    db = engine.SketchDB(sk, 16, 14)
    wi, wj, wd = engine.knn_from_sketches(db, kmers, tbl, 7, dist_col=1, method="tiles")
    bounds = engine.shard_bounds(db.n, 0, 4)
    parts = [engine.knn_candidates(db, kmers, tbl, 7, dist_col=1, q_begin=bounds[i], q_end=bounds[i + 1], cap=1000)
             for i in (2, 0, 3, 1)]                     # cap 1000: the capacity retry is exercised
    keys = torch.cat([p[0] for p in parts])
    vals = torch.cat([p[1] for p in parts])
    gi, gj, gd = engine.knn_select(keys, vals, db.n, 7)
    assert torch.equal(gi, wi) and torch.equal(gj, wj) and torch.equal(gd, wd)
    db.close()


def test_knn_from_tiles_at_50000_without_any_matrix():
    """n = 50 000 (beyond the 46 340 where the square stops fitting the default budget): the tile path
    equals the band-by-band path -- which computes both triangles -- neighbour for neighbour, from a
    few dozen candidates per sample instead of n."""
    import torch
    from poppunk_amd import engine, synth
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    t = synth.make_sketches_device(50000, kmers, seed=50, device="cuda:0")
    db = engine.SketchDB(t, 16, 14)
    del t
    tbl = synth.random_match_table(kmers)
    info = {}
    ti, tj, td = engine.knn_from_sketches(db, kmers, tbl, 5, method="tiles", info=info)
    bi, bj, bd = engine.knn_from_sketches(db, kmers, tbl, 5, method="bands")
    assert torch.equal(ti, bi) and torch.equal(td, bd) and torch.equal(tj, bj)
    assert info["candidates"] < 50000 * 1000         # vs 2.5e9 pair-roles
    print("50 000 samples, k = 5: %d candidates (%.1f per sample)" % (info["candidates"], info["candidates"] / 50000))
    db.close()
    torch.cuda.empty_cache()


def test_prune_distance_matrix_golden_and_random(tmp_path, capsys):
    """qc.prune_distance_matrix (ppk_prune_long) against the reference's own outputs
    (tests/golden/prune.json) and, on a larger matrix, against the oracle; the .pkl/.npy pair is
    written like the reference does."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prune.json")))
    for k, c in enumerate(g["self"]):
        dist = np.asarray(c["dist"], dtype=np.float32)
        names, new = qc.prune_distance_matrix(c["names"], c["remove"], dist, str(tmp_path / ("p%d" % k)))
        assert list(names) == c["new_names"]
        assert np.array_equal(np.asarray(new).reshape(-1, 2),
                              np.asarray(c["new_dist"], dtype=np.float32).reshape(-1, 2))
        assert os.path.exists(str(tmp_path / ("p%d.pkl" % k)))
    assert "Couldn't find not_in_db in database" in capsys.readouterr().err
    rng = np.random.Generator(np.random.PCG64(5))
    n = 1500
    dist = rng.random((n * (n - 1) // 2, 2)).astype(np.float32)
    names = ["g%d" % i for i in range(n)]
    gone = sorted(rng.choice(n, size=137, replace=False))
    new_names, new = qc.prune_distance_matrix(names, [names[i] for i in gone], dist, None)
    keep = [i for i in range(n) if i not in set(gone)]
    assert new_names == [names[i] for i in keep]
    assert np.array_equal(new, oracle.prune_long(dist, n, keep))
    # any dtype / layout goes in and the caller's dtype comes out, as in the reference (newDistMat takes
    # distMat.dtype, PopPUNK/qc.py:75): round-3 advisor finding
    names64, new64 = qc.prune_distance_matrix(names, [names[i] for i in gone], np.asfortranarray(dist.astype(np.float64)), None)
    assert names64 == new_names and new64.dtype == np.float64 and np.array_equal(new64, new.astype(np.float64))


def test_prune_on_resident_buffers():
    import torch
    rng = np.random.Generator(np.random.PCG64(6))
    n = 700
    dist = rng.random((n * (n - 1) // 2, 2)).astype(np.float32)
    keep = np.sort(rng.choice(n, size=412, replace=False))
    d = torch.as_tensor(dist, device="cuda")
    got = engine.prune_long_dev(d, n, keep).cpu().numpy()
    assert np.array_equal(got, oracle.prune_long(dist, n, keep))
    one = engine.prune_long_dev(d[:, 1].contiguous(), n, keep).cpu().numpy()    # a single column
    assert np.array_equal(one, oracle.prune_long(dist[:, 1], n, keep))
    with pytest.raises(RuntimeError):
        engine.prune_long_dev(d, n, keep[::-1].copy())
    n_ref, n_qry = 300, 90
    qr = rng.random((n_ref * n_qry, 2)).astype(np.float32)
    kq = np.sort(rng.choice(n_qry, size=55, replace=False))
    got = engine.prune_query_rows_dev(torch.as_tensor(qr, device="cuda"), n_ref, kq).cpu().numpy()
    assert np.array_equal(got, qr.reshape(n_qry, n_ref, 2)[kq].reshape(-1, 2))


@pytest.mark.parametrize("knn", [1, 5, 12])
def test_neighbours_from_tiles_in_pieces_with_a_short_candidate_list(ppk_option, knn):
    """Large jobs run piece by piece and cut the candidate list back to the best knn per sample whenever it
    is half full (a million genomes would otherwise emit ~10^10 candidates).  Forced at 1 500 genomes with a
    tiny tile budget and a short list: many selections, pieces that do not fit and run again -- the result is
    the single-pass result, which is the oracle's."""
    from poppunk_amd import engine
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(1500, kmers, cluster_size=40, seed=19)
    tbl = synth.random_match_table(kmers)
    db = engine.SketchDB(sk, 16, 14)
    info = {}
    wi, wj, wd = (x.cpu().numpy() for x in engine.knn_from_sketches(db, kmers, tbl, knn, method="tiles", info=info))
    d, _ = engine.dist(db, None, kmers, tbl)
    oi, oj, od = oracle.knn(oracle.long_to_square(d.cpu().numpy()[:, 0]), knn)
    assert np.array_equal(wi, oi) and np.array_equal(wj, oj) and np.array_equal(wd, od)
    single = info["candidates"]
    for tiles, room in ((12, 40000), (6, 1), (48, 1 << 19)):
        ppk_option("launch_tiles", tiles)
        ppk_option("knn_list", room)
        gi, gj, gd = (x.cpu().numpy() for x in engine.knn_from_sketches(db, kmers, tbl, knn, method="tiles", info=info))
        assert np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(gd, od), (tiles, room)
    assert single > 1500 * knn
    db.close()


def test_candidate_bands_beyond_one_dispatch(ppk_option):
    """The multi-process form (candidates per band, then one selection) when a band holds more tiles than one
    launch may: ppk_launch_dist appends the pieces' candidates through the shared counter."""
    import torch
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(1300, kmers, cluster_size=26, seed=23)
    tbl = synth.random_match_table(kmers)
    db = engine.SketchDB(sk, 16, 14)
    wi, wj, wd = (x.cpu().numpy() for x in engine.knn_from_sketches(db, kmers, tbl, 6, dist_col=1, method="tiles"))
    ppk_option("launch_tiles", 10)
    parts = [engine.knn_candidates(db, kmers, tbl, 6, dist_col=1, q_begin=a, q_end=b) for a, b in ((0, 500), (500, 1300))]
    keys = torch.cat([p[0] for p in parts])
    vals = torch.cat([p[1] for p in parts])
    gi, gj, gd = (x.cpu().numpy() for x in engine.knn_select(keys, vals, 1300, 6))
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gd, wd)
    db.close()


def test_neighbours_from_tiles_at_the_default_sketch_size():
    """PopPUNK's default sketch size (s = 9 984: 156 blocks per k, 14-bit counts, three-dword count registers):
    neighbours from the tiles, self and ref x query, against a stable sort of the oracle's distances."""
    kmers = np.asarray([13, 17, 21, 25, 29], dtype=np.int32)
    sk, _ = synth.make_sketches(700, kmers, sketchsize64=156, bbits=14, cluster_size=35, seed=5)
    tbl = synth.random_match_table(kmers)
    db = engine.SketchDB(sk, 156, 14)
    want, _ = oracle.query(sk, None, kmers, 156, 14, tbl, threads=8)
    for knn, col in ((4, 0), (11, 1)):
        wi, wj, wd = oracle.knn(oracle.long_to_square(want[:, col]), knn)
        gi, gj, gd = (x.cpu().numpy() for x in engine.knn_from_sketches(db, kmers, tbl, knn, dist_col=col, method="tiles"))
        assert np.array_equal(gj, wj) and np.abs(gd - wd).max() <= 1e-6
    rdb, qdb = engine.SketchDB(sk[:500], 156, 14), engine.SketchDB(sk[500:], 156, 14)
    rq, _ = oracle.query(sk[:500], sk[500:], kmers, 156, 14, tbl, threads=8)
    rect = rq[:, 0].reshape(200, 500)
    gi, gj, gd = (x.cpu().numpy() for x in engine.knn_ref_query(rdb, qdb, kmers, tbl, 3))
    gj = gj.reshape(700, 3)
    for r in (0, 17, 499):
        assert gj[r].tolist() == (np.argsort(rect[:, r], kind="stable")[:3] + 500).tolist()
    for q in (0, 99, 199):
        assert gj[500 + q].tolist() == np.argsort(rect[q], kind="stable")[:3].tolist()
    for x in (db, rdb, qdb):
        x.close()
