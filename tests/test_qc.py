"""Distance QC either side of the distance call (PopPUNK/qc.py:238-369,:419-468): qcDistMat, prune_edges,
autoDistFind against what the reference's own functions returned (tests/golden/qc.json, made by
tests/golden/make_golden.py golden_qc), and the device edge lists against numpy + the oracle."""
import json
import os

import numpy as np
import pytest

from poppunk_amd import qc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "qc.json")) as f:
        return json.load(f)


def test_prune_edges_golden(gold):
    assert len(gold["prune_edges"]) >= 6
    for c in gold["prune_edges"]:
        before = None if c["failed_before"] is None else set(c["failed_before"])
        for edges in ([tuple(e) for e in c["edges"]], np.asarray(c["edges"], dtype=np.int64).reshape(-1, 2)):
            got = qc.prune_edges(edges, c["query_start"], failed=None if before is None else set(before),
                                 min_count=c["min_count"], allow_ref_ref=c["allow_ref_ref"])
            assert sorted(got) == c["failed"]


def test_autoDistFind_golden(gold, capfd):
    mats = np.load(os.path.join(HERE, "golden", "qc_autodist.npz"))
    for c in gold["autoDistFind"]:
        d = mats[c["dist"].split(":")[1]]
        max_pi, max_a = qc.autoDistFind(d, {"x": c["x"], "r": c["r"]})
        assert float(max_pi) == c["max_pi"] and float(max_a) == c["max_a"]
    assert "Detecting maximum distance cutoffs using x = " in capfd.readouterr().err
    # nothing stands out: the column maxima, and a message each
    flat = np.linspace(0.001, 0.002, 4000, dtype=np.float32).reshape(-1, 2)
    assert qc.autoDistFind(flat, {"x": 0.2, "r": 10}) == (flat[:, 0].max(), flat[:, 1].max())
    err = capfd.readouterr().err
    assert "No outlier detected in core distance" in err and "No outlier detected in accessory distance" in err


@pytest.mark.gpu
def test_qcDistMat_golden(gold, capfd):
    assert len(gold["qcDistMat"]) >= 7
    for c in gold["qcDistMat"]:
        d = np.asarray(c["dist"], dtype=np.float32)
        kept, failed = qc.qcDistMat(d, c["refs"], c["queries"], "unused_db", c["qc_dict"])
        assert kept == c["retained"]
        assert failed == c["failed"]
    err = capfd.readouterr().err
    assert "Running QC on distances\n" in err and "Using cutoff for proportion of zero distances: " in err


@pytest.mark.gpu
@pytest.mark.parametrize("n_ref,n_qry", [(700, 0), (300, 211)])
def test_qc_edge_lists_equal_masks_and_oracle_tuples(n_ref, n_qry):
    """Both lists of one call equal the reference's construction -- numpy masks as 0/1 rows, then
    generateTuples(rows, 0, ...) (PopPUNK/qc.py:331-337,:348-354) with the oracle's generate_tuples -- also
    when they do not fit the first buffer (more than 65 536 edges: the parked result is fetched)."""
    from oracle import oracle
    rng = np.random.Generator(np.random.PCG64(77))
    rows = n_ref * (n_ref - 1) // 2 if n_qry == 0 else n_ref * n_qry
    d = np.stack([rng.uniform(0, 0.04, rows), rng.uniform(0, 0.6, rows)], axis=1).astype(np.float32)
    d[rng.choice(rows, rows // 50, replace=False), 0] = 0.0
    d[rng.choice(rows, rows // 70, replace=False), 1] = 0.0
    for max_pi, max_a in ((0.039, 0.59), (0.03, 0.45), (1.0, 1.0)):
        long_rows = np.where((d[:, 0] > max_pi) | (d[:, 1] > max_a), 0, 1).astype(np.int32)
        zero_rows = np.where((d[:, 0] == 0) | (d[:, 1] == 0), 0, 1).astype(np.int32)
        want_long = oracle.generate_tuples(long_rows, 0, self=n_qry == 0, num_ref=n_ref, int_offset=0)
        want_zero = oracle.generate_tuples(zero_rows, 0, self=n_qry == 0, num_ref=n_ref, int_offset=0)
        got_long, got_zero = qc.qc_edge_lists(d, 0 if n_qry == 0 else n_ref, max_pi, max_a)
        assert np.array_equal(got_long, np.asarray(want_long).reshape(-1, 2))
        assert np.array_equal(got_zero, np.asarray(want_zero).reshape(-1, 2))
        only_long, none = qc.qc_edge_lists(d, 0 if n_qry == 0 else n_ref, max_pi, max_a, zeros=False)
        assert none is None and np.array_equal(only_long, got_long)
    assert len(got_zero) > 1000 and len(got_long) == 0
    # any dtype / layout, as the reference's numpy comparisons take them (round-3 advisor finding)
    as64, _ = qc.qc_edge_lists(d.astype(np.float64), 0 if n_qry == 0 else n_ref, max_pi, max_a)
    assert np.array_equal(as64, got_long)
    strided, _ = qc.qc_edge_lists(np.asfortranarray(d), 0 if n_qry == 0 else n_ref, max_pi, max_a)
    assert np.array_equal(strided, got_long)
    with pytest.raises(TypeError):
        qc.qc_edge_lists(d[:, 0], 0, 0.1, 0.1)
