"""SURVEY.md 8f rank 2 on a real MI355X: long <-> square transforms and kNN vs scipy/numpy."""
import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import poppunk_refine, pp_sketchlib

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


@pytest.mark.parametrize("n,k", [(5, 2), (64, 3), (300, 10), (300, 1), (4, 7)])
def test_knn(n, k):
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
        gi, gj, gd = engine.knn_from_sketches(db, kmers, tbl, k, dist_col=col, band_items=500 * 128)
        assert np.array_equal(gi.cpu().numpy(), wi)
        assert np.array_equal(gd.cpu().numpy(), wd)
        # ties between equal distances resolve by column index in both
        assert np.array_equal(gj.cpu().numpy(), wj)
    db.close()
