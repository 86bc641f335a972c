"""The reference's one numeric test at the queryDatabase boundary, test/test-update-gpu.py:85-90 and
:121-126: the distances of a database grown with `poppunk_assign --update-db` (stored ref-ref distances
+ query-ref + query-query, merged by update_distance_matrices, PopPUNK/utils.py:357-408, and brought
back to long form by squareToLong, PopPUNK/network.py:2133-2134) equal those of the database built from
all genomes at once.  The reference asks for R^2 >= 0.99 between the two (its CUDA path is fp32
fast-math); here every pair's fit is the same arithmetic whichever call computes it, so the bar is
bit-for-bit, plus the oracle within 1e-6.
"""
import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import pp_sketchlib, sketchdb, synth
from poppunk_amd.utils import iterDistRows, update_distance_matrices

pytestmark = pytest.mark.gpu

KMERS = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
TOL = 1e-6


def _grow(tmp_path, n_ref, n_qry, seed, second_batch=0):
    n = n_ref + n_qry + second_batch
    sk, _ = synth.make_sketches(n, KMERS, cluster_size=40, seed=seed)
    tbl = synth.random_match_table(KMERS)
    names = ["g%05d" % i for i in range(n)]
    full = str(tmp_path / "batch_all")            # the database of everything (batch12 / batch123)
    refdb = str(tmp_path / "batch1")              # the first batch alone
    qrydb = str(tmp_path / "query")               # what poppunk_assign sketches for the new genomes
    sketchdb.save_npz(full, names, KMERS, sk, 16, 14, random_table=tbl)
    sketchdb.save_npz(refdb, names[:n_ref], KMERS, sk[:n_ref], 16, 14, random_table=tbl)
    sketchdb.save_npz(qrydb, names[n_ref:], KMERS, sk[n_ref:], 16, 14, random_table=tbl)
    return sk, tbl, names, full, refdb, qrydb


def _update(refdb, qrydb, rNames, qNames, rr, klist):
    """One --update-db step: the stored rr, fresh qr and qq, merged and brought back to long form."""
    qr = pp_sketchlib.queryDatabase(refdb, qrydb, rNames, qNames, klist, True, False, 1, True, 0)
    qq = pp_sketchlib.queryDatabase(qrydb, qrydb, qNames, qNames, klist, True, False, 1, True, 0)
    assert qr.shape == (len(rNames) * len(qNames), 2) and qq.shape == (len(qNames) * (len(qNames) - 1) // 2, 2)
    labels, core, acc = update_distance_matrices(rNames, rr, qNames, qr, qq, threads=4)
    assert labels == rNames + qNames
    n = len(labels)
    assert core.shape == acc.shape == (n, n) and core.dtype == np.float32
    assert np.array_equal(core, core.T) and not core.diagonal().any()
    long = np.hstack((pp_sketchlib.squareToLong(core, 4).reshape(-1, 1),
                      pp_sketchlib.squareToLong(acc, 4).reshape(-1, 1)))
    return long, qr, qq


@pytest.mark.parametrize("n_ref,n_qry", [(130, 70), (64, 1), (2, 300), (1000, 257)])
def test_updated_database_distances_equal_the_from_scratch_ones(tmp_path, n_ref, n_qry):
    sk, tbl, names, full, refdb, qrydb = _grow(tmp_path, n_ref, n_qry, seed=n_ref + n_qry)
    klist = KMERS.tolist()
    pp_sketchlib.clear_cache()
    rNames, qNames = names[:n_ref], names[n_ref:]
    rr = pp_sketchlib.queryDatabase(refdb, refdb, rNames, rNames, klist, True, False, 1, True, 0)
    X2, qr, qq = _update(refdb, qrydb, rNames, qNames, rr, klist)
    X1 = pp_sketchlib.queryDatabase(full, full, names, names, klist, True, False, 1, True, 0)
    assert X1.shape == X2.shape == (len(names) * (len(names) - 1) // 2, 2)
    assert np.array_equal(X1, X2)                                  # bit for bit
    want, _ = oracle.query(sk, None, KMERS, 16, 14, tbl, threads=8)
    assert np.abs(X1 - want).max() <= TOL
    # the three pieces against the oracle on their own, and the row order of the pickle that goes with them
    assert np.abs(qr - oracle.query(sk[:n_ref], sk[n_ref:], KMERS, 16, 14, tbl, threads=8)[0]).max() <= TOL
    if n_qry > 1:
        assert np.abs(qq - oracle.query(sk[n_ref:], None, KMERS, 16, 14, tbl, threads=8)[0]).max() <= TOL
    rows = list(iterDistRows(names, names, self=True))
    assert len(rows) == len(X1) and rows[0] == (names[1], names[0]) and rows[-1] == (names[-1], names[-2])
    # no queries: the plain longToSquare leg (utils.py:392-396)
    labels, core, acc = update_distance_matrices(rNames, rr)
    assert labels == rNames and np.array_equal(core, oracle.long_to_square(rr[:, 0]))
    assert np.array_equal(acc, oracle.long_to_square(rr[:, 1]))
    pp_sketchlib.clear_cache()


def test_two_successive_updates_like_test_update_gpu(tmp_path):
    """batch1 -> (+ batch2) -> (+ batch3), test/test-update-gpu.py:70-126: after the second update the
    stored matrix is the first update's result."""
    n1, n2, n3 = 300, 150, 77
    sk, tbl, names, full, _, _ = _grow(tmp_path, n1, n2, seed=5, second_batch=n3)
    klist = KMERS.tolist()
    pp_sketchlib.clear_cache()
    b1, b2, b3 = names[:n1], names[n1:n1 + n2], names[n1 + n2:]
    rr = pp_sketchlib.queryDatabase(full, full, b1, b1, klist, True, False, 1, True, 0)
    # every batch is read out of the full file here (queryDatabase takes any name subset of a database)
    X12, _, _ = _update(full, full, b1, b2, rr, klist)
    assert np.array_equal(X12, pp_sketchlib.queryDatabase(full, full, b1 + b2, b1 + b2, klist, True, False, 1, True, 0))
    X123, _, _ = _update(full, full, b1 + b2, b3, X12, klist)
    X1 = pp_sketchlib.queryDatabase(full, full, names, names, klist, True, False, 1, True, 0)
    assert np.array_equal(X123, X1)
    assert np.abs(X1 - oracle.query(sk, None, KMERS, 16, 14, tbl, threads=8)[0]).max() <= TOL
    pp_sketchlib.clear_cache()


def test_update_at_config_size_10000_refs_plus_2000_queries(tmp_path):
    """BASELINE config 3's database grown by 2 000 genomes: 72 M rows, both routes, bit for bit; the
    query-ref block (poppunk_assign's own product) against the oracle in full."""
    n_ref, n_qry = 10000, 2000
    sk, tbl, names, full, refdb, qrydb = _grow(tmp_path, n_ref, n_qry, seed=20260928)
    klist = KMERS.tolist()
    pp_sketchlib.clear_cache()
    rNames, qNames = names[:n_ref], names[n_ref:]
    rr = pp_sketchlib.queryDatabase(refdb, refdb, rNames, rNames, klist, True, False, 1, True, 0)
    X2, qr, _ = _update(refdb, qrydb, rNames, qNames, rr, klist)
    X1 = pp_sketchlib.queryDatabase(full, full, names, names, klist, True, False, 1, True, 0)
    assert X1.shape == (12000 * 11999 // 2, 2)
    assert np.array_equal(X1, X2)
    want = oracle.query(sk[:n_ref], sk[n_ref:], KMERS, 16, 14, tbl, threads=16)[0]
    assert np.abs(qr - want).max() <= TOL
    pp_sketchlib.clear_cache()
