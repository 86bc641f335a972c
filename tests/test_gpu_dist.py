"""Kernel 1 parity on a real MI355X: HIP path (through the C ABI) vs the CPU oracle.

Integer half (match counts): bit-exact.  Regressed distances: |d| <= 1e-6 (the
tolerance BASELINE.json's north_star states; fp64 regression on both sides, the
only difference is the device libm's log/exp vs glibc's).
"""
import os

import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import _lib, engine, pp_sketchlib, synth

pytestmark = pytest.mark.gpu

KMERS = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
TOL = 1e-6


@pytest.fixture(scope="module")
def sk300():
    sk, member = synth.make_sketches(300, KMERS, cluster_size=30)
    return sk, member


@pytest.fixture(scope="module")
def tbl1():
    return synth.random_match_table(KMERS)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 129, 300])
def test_counts_self_bit_exact(sk300, n):
    sk = sk300[0][:n]
    got, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, counts=True)
    want = oracle.match_counts(sk, None, 16, 14, threads=4)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nr,nq", [(1, 1), (5, 3), (64, 64), (65, 17), (130, 70), (257, 9)])
def test_counts_ref_query_bit_exact(sk300, nr, nq):
    sk = sk300[0]
    ref, qry = sk[:nr], sk[300 - nq:]
    got, _ = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, counts=True)
    want = oracle.match_counts(ref, qry, 16, 14, threads=4)
    assert np.array_equal(got, want)


def test_identical_and_disjoint_sketches():
    rng = np.random.Generator(np.random.PCG64(7))
    bins = rng.integers(0, 1 << 14, size=(3, 5, 1024), dtype=np.uint16)
    bins[1] = bins[0]                                  # identical pair -> all 1024 bins match
    bins[2] = (bins[0] + 1) & 0x3FFF                   # no bin equal
    sk = synth.bitslice(bins, 14)
    got, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, counts=True)
    assert np.array_equal(got[0], np.full(5, 1024))     # (0,1)
    assert np.array_equal(got[1], np.zeros(5))          # (0,2)
    d, failed = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, None, random_correct=False)
    assert np.array_equal(d[0], [0.0, 0.0])             # identical genomes: J=1 at every k
    assert failed == 2 and np.array_equal(d[1:], np.zeros((2, 2)))


@pytest.mark.parametrize("ksplit", [0, 640])
def test_distances_self(sk300, tbl1, ppk_option, ksplit):
    """tile epilogue (ksplit 0) and the small-job k-split path (default threshold)"""
    sk = sk300[0]
    ppk_option("ksplit", ksplit)
    got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, tbl1, threads=4)
    assert gf == wf
    assert np.abs(got - want).max() <= TOL
    assert got.dtype == np.float32 and got.flags["C_CONTIGUOUS"]


def test_distances_ref_query(sk300, tbl1):
    sk = sk300[0]
    ref, qry = sk[:171], sk[171:]
    got, gf = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, tbl1)
    want, wf = oracle.query(ref, qry, KMERS, 16, 14, tbl1, threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL


def test_no_random_correction(sk300):
    sk = sk300[0][:120]
    got, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, None, random_correct=False)
    want, _ = oracle.query(sk, None, KMERS, 16, 14, None, random_correct=False, threads=4)
    assert np.abs(got - want).max() <= TOL


def test_jaccard_mode(sk300, tbl1):
    sk = sk300[0][:150]
    for rc in (True, False):
        got, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1, random_correct=rc,
                                           jaccard=True)
        want, _ = oracle.query(sk, None, KMERS, 16, 14, tbl1, random_correct=rc, jaccard=True,
                               threads=4)
        assert got.shape == (150 * 149 // 2, 5)
        assert np.abs(got - want).max() <= 1e-7


def test_multi_cluster_random_table(sk300):
    sk, member = sk300
    rng = np.random.Generator(np.random.PCG64(11))
    n_clu = 3
    tbl = synth.random_match_table(KMERS, n_clu=n_clu)
    tbl = (tbl * rng.uniform(0.5, 3.0, size=tbl.shape)).astype(np.float32)
    clu = (member % n_clu).astype(np.uint16)
    got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl, clu)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, tbl, clu, clu, threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL
    ref, qry = sk[:200], sk[200:]
    got, gf = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, tbl, clu[:200], clu[200:])
    want, wf = oracle.query(ref, qry, KMERS, 16, 14, tbl, clu[:200], clu[200:], threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL


def test_unrelated_clusters_failed_fits(tbl1):
    """Different-cluster pairs match only by chance -> < 2 usable k -> (0,0), counted."""
    sk, _ = synth.make_sketches(200, KMERS, cluster_size=20, related=False)
    got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, tbl1, threads=4)
    assert gf == wf and gf > 0
    assert np.abs(got - want).max() <= TOL


def test_bands_concatenate_to_whole(sk300, tbl1):
    """The multi-GPU split: any band partition of the query axis reproduces the matrix."""
    import torch
    sk = sk300[0]
    db = engine.SketchDB(sk, 16, 14, device=0)
    whole, _ = engine.dist(db, None, KMERS, tbl1)
    for parts in (2, 3, 8):
        b = engine.band_split(300, 0, parts)
        pieces = [engine.dist(db, None, KMERS, tbl1, q_begin=b[i], q_end=b[i + 1])[0]
                  for i in range(parts)]
        assert torch.equal(torch.cat(pieces), whole)
    # un-aligned band edges are allowed as well
    pieces = [engine.dist(db, None, KMERS, tbl1, q_begin=a, q_end=e)[0]
              for a, e in ((0, 37), (37, 38), (38, 201), (201, 300))]
    assert torch.equal(torch.cat(pieces), whole)
    want, _ = oracle.query(sk, None, KMERS, 16, 14, tbl1, threads=4)
    assert np.abs(whole.cpu().numpy() - want).max() <= TOL
    db.close()


def test_host_query_multi_device_split(sk300, tbl1):
    """ppk_query's band loop with a device list (the same GPU twice on a 1-GPU box)."""
    sk = sk300[0][:200]
    one, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1, devices=(0,))
    two, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1, devices=(0, 0, 0))
    assert np.array_equal(one, two)


def test_config1_real_sketch_and_default_sketch_size(golden_dir):
    """BASELINE config 1 shape: s = 9984 (sketchsize64 156), k = 13..28 step 3 -> 6 k-mer
    lengths x 14-bit counts = 84 bits -> the 128-bit packed path.  The real sketch of
    test/json_sketch.txt must match itself in all 9984 bins at every k -> (0, 0)."""
    import os
    z = np.load(os.path.join(golden_dir, "json_sketch.npz"))
    kmers = z["kmers"]
    s64, bbits = int(z["sketchsize64"]), int(z["bbits"])
    real = z["sketch"][None]                                   # [1, 6, 2184]
    syn, _ = synth.make_sketches(28, kmers, sketchsize64=s64, bbits=bbits, cluster_size=7, seed=5)
    sk = np.concatenate([real, real, syn])                     # sample 0 == sample 1
    tbl = synth.random_match_table(kmers)
    counts, _ = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, counts=True)
    assert np.array_equal(counts, oracle.match_counts(sk, None, s64, bbits, threads=4))
    assert np.array_equal(counts[0], np.full(6, 9984))
    got, gf = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, tbl)
    want, wf = oracle.query(sk, None, kmers, s64, bbits, tbl, threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL
    assert np.array_equal(got[0], [0.0, 0.0])


@pytest.mark.parametrize("s64,bbits", [(4, 8), (3, 14), (20, 10), (300, 14)])
def test_other_sketch_shapes(s64, bbits):
    """Generic-bbits kernel, odd block counts (chunk tail), and 64*s64 >= 2^bbits, where the
    collision adjustment of row a4 is non-zero."""
    kmers = np.asarray([13, 17, 21, 25], dtype=np.int32)
    sk, _ = synth.make_sketches(90, kmers, sketchsize64=s64, bbits=bbits, cluster_size=9, seed=3)
    tbl = synth.random_match_table(kmers)
    counts, _ = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, counts=True)
    assert np.array_equal(counts, oracle.match_counts(sk, None, s64, bbits, threads=4))
    got, gf = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, tbl)
    want, wf = oracle.query(sk, None, kmers, s64, bbits, tbl, threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL


@pytest.mark.parametrize("s64,bbits", [(4, 8), (20, 10), (300, 14)])
def test_ext_collision_gate_both_ways(ppk_option, s64, bbits):
    """[EXT] a4: with expected = nbins >> bbits > 0 the two readings of
    calc_intersize's gate give different numbers; kernel and oracle flip together (one switch each)
    and agree under both -- tile kernel (bbits 14), generic kernel, k-split and regression pass."""
    kmers = np.asarray([13, 17, 21, 25], dtype=np.int32)
    assert (64 * s64) >> bbits > 0
    sk, _ = synth.make_sketches(90, kmers, sketchsize64=s64, bbits=bbits, cluster_size=9, seed=3)
    tbl = synth.random_match_table(kmers)
    res = {}
    try:
        for gate in (0, 1):
            ppk_option("ext_collision_adjust", gate)
            oracle.set_ext(collision_adjust=gate)
            want, wf = oracle.query(sk, None, kmers, s64, bbits, tbl, threads=4)
            wj, _ = oracle.query(sk, None, kmers, s64, bbits, tbl, jaccard=True, threads=4)
            for ks in (0, 640):
                ppk_option("ksplit", ks)
                got, gf = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, tbl)
                assert gf == wf and np.abs(got - want).max() <= TOL
            jac, _ = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, tbl, jaccard=True)
            assert np.array_equal(jac, wj)
            res[gate] = got
    finally:
        oracle.set_ext(0, 0)
    assert np.abs(res[0] - res[1]).max() > 1e-5        # the gate matters at these sketch sizes


def test_ext_fit_skip_both_ways(ppk_option):
    """[EXT] a6: truncate at (default) or skip the k-mer lengths whose J < 5/s."""
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    # distant but related samples: J falls through the 5/1024 floor at the longer k
    rng = np.random.Generator(np.random.PCG64(8))
    bins = rng.integers(0, 1 << 14, size=(1, 5, 1024), dtype=np.uint16).repeat(300, axis=0)
    keep_p = np.asarray([0.25, 0.1, 0.05, 0.1, 0.05])          # J ~ 0.14, 0.05, 0.026, 0.05, 0.026 ...
    redraw = rng.random(bins.shape) > keep_p[None, :, None]
    fresh = rng.integers(0, 1 << 14, size=bins.shape, dtype=np.uint16)
    bins[redraw] = fresh[redraw]
    sk = synth.bitslice(bins, 14)
    tbl = synth.random_match_table(kmers)
    res = {}
    try:
        for skip in (0, 1):
            ppk_option("ext_fit_skip", skip)
            oracle.set_ext(fit_skip=skip)
            want, wf = oracle.query(sk, None, kmers, 16, 14, tbl, threads=4)
            for ks in (0, 640):
                ppk_option("ksplit", ks)
                got, gf = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl)
                assert gf == wf and np.abs(got - want).max() <= TOL
            res[skip] = (got, gf)
    finally:
        oracle.set_ext(0, 0)
    assert res[1][1] < res[0][1]                               # skipping rescues fits truncation fails
    assert np.abs(res[0][0] - res[1][0]).max() > 1e-4


def test_two_kmers_minimum():
    kmers = np.asarray([13, 29], dtype=np.int32)
    sk, _ = synth.make_sketches(70, kmers, cluster_size=10, seed=9)
    got, gf = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, None, random_correct=False)
    want, wf = oracle.query(sk, None, kmers, 16, 14, None, random_correct=False, threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL


# ---- BASELINE sizes: size-independent properties ---------------------------------------

@pytest.fixture(scope="module")
def big():
    sk, _ = synth.make_sketches(10000, KMERS)
    return sk


def test_full_size_properties_10k(big, tbl1):
    """configs[2] workload (10 000 self, 49 995 000 pairs).  The oracle cannot cover it in
    seconds, so: (1) a ref x query run of a sample against the whole DB must equal the
    matching rows of the self run (two different tilings/row orders of the same pairs);
    (2) an oracle spot check on rows of scattered queries; (3) checksum of band results
    equals the checksum of the whole."""
    import torch
    db = engine.SketchDB(big, 16, 14, device=0)
    n = 10000
    whole, nf = engine.dist(db, None, KMERS, tbl1)
    assert whole.shape == (n * (n - 1) // 2, 2)
    assert torch.isfinite(whole).all()
    assert float(whole.min()) >= 0.0 and float(whole.max()) <= 1.0

    qsel = np.asarray([0, 1, 63, 64, 4999, 9936, 9998], dtype=np.int64)
    qdb = engine.SketchDB(big[qsel], 16, 14, device=0)
    sub, _ = engine.dist(db, qdb, KMERS, tbl1)                  # row = qi*n + r
    sub = sub.cpu().numpy().reshape(len(qsel), n, 2)
    w = whole.cpu().numpy()
    for a, q in enumerate(qsel):
        start = q * n - q * (q + 1) // 2
        assert np.array_equal(sub[a, q + 1:], w[start:start + n - 1 - q])    # pairs (q, r>q)
    want, _ = oracle.query(big, big[qsel[:3]], KMERS, 16, 14, tbl1, threads=8)
    assert np.abs(sub[:3].reshape(-1, 2) - want).max() <= TOL

    b = engine.band_split(n, 0, 8)
    s = 0.0
    for i in range(8):
        piece, _ = engine.dist(db, None, KMERS, tbl1, q_begin=b[i], q_end=b[i + 1])
        s += float(piece.double().sum())
    assert abs(s - float(whole.double().sum())) <= 1e-6 * abs(s)
    db.close()
    qdb.close()


def test_row_index_math_beyond_32_bits(big, tbl1):
    """n = 70 000 self -> 2.45e9 rows (> 2^31): a band deep in the matrix must land on the right
    64-bit row offsets.  The database is the 10k set tiled 7x (duplicates have J = 1)."""
    import torch
    n = 70000
    sk = np.tile(big, (7, 1, 1))
    db = engine.SketchDB(sk, 16, 14, device=0)
    qb, qe = 60000, 60064
    band, _ = engine.dist(db, None, KMERS, tbl1, q_begin=qb, q_end=qe)
    rows = engine.rows_in_band(n, 0, qb, qe)
    assert band.shape[0] == rows == sum(n - 1 - q for q in range(qb, qe))
    want, _ = oracle.query(sk, sk[qb:qe], KMERS, 16, 14, tbl1, threads=8)      # row = qi*n + r
    want = want.reshape(qe - qb, n, 2)
    got = band.cpu().numpy()
    off = 0
    for a, q in enumerate(range(qb, qe)):
        cnt = n - 1 - q
        assert np.abs(got[off:off + cnt] - want[a, q + 1:]).max() <= TOL
        off += cnt
    # fused edges of a band carry 64-bit-safe (i, j): identical genomes (core 0) are 10 000 apart
    eb, ee = 20032, 20096
    e, _ = engine.dist_edges(db, None, KMERS, tbl1, slope=0, x_max=1e-9, y_max=0.0, q_begin=eb, q_end=ee)
    e = e.cpu().numpy()
    dup = {(q, q + 10000 * m) for q in range(eb, ee) for m in range(1, 7) if q + 10000 * m < n}
    assert len(dup) == 64 * 4 and dup <= set(map(tuple, e.tolist()))
    assert np.all(e[:, 0] >= eb) and np.all(e[:, 0] < ee) and np.all(e[:, 1] > e[:, 0]) and np.all(e[:, 1] < n)
    db.close()


def test_config4_shape_50k_queries_x_10k_refs(big, tbl1):
    """BASELINE configs[3] at full size: 50 000 queries x 10 000 refs (5e8 pairs), refs resident.
    The queries are the 10k set tiled 5x, so the result must be 5 identical 10k x 10k blocks
    (row = q*n_ref + r), each block symmetric with a zero diagonal and equal, on its upper
    triangle, to the self run of the 10k set -- three different tilings of the same pairs."""
    import torch
    n = 10000
    ref = engine.SketchDB(big, 16, 14, device=0)
    qry = engine.SketchDB(np.tile(big, (5, 1, 1)), 16, 14, device=0)
    out, nf = engine.dist(ref, qry, KMERS, tbl1)
    assert out.shape == (5 * n * n, 2) and int(nf.item()) == 0
    blocks = out.view(5, n, n, 2)
    for b in range(1, 5):
        assert torch.equal(blocks[b], blocks[0])
    sq = blocks[0]
    assert torch.equal(sq, sq.transpose(0, 1).contiguous())
    assert float(sq.diagonal(dim1=0, dim2=1).abs().max()) == 0.0
    self_run, _ = engine.dist(ref, None, KMERS, tbl1)
    iu = torch.triu_indices(n, n, offset=1, device=out.device)
    assert torch.equal(sq[iu[0], iu[1]], self_run)
    # and the long <-> square kernels agree with that
    for col in (0, 1):
        assert torch.equal(engine.long_to_square_dev(self_run, col, n), sq[:, :, col].contiguous())
    ref.close()
    qry.close()


def test_config5_shape_100k_self_fused_boundary_band(big, tbl1):
    """BASELINE configs[4] at full size: 100 000 genomes self, distances never materialised,
    boundary applied in-kernel, edge list compacted on the device -- for the band one of 8 GPUs
    owns.  The 100k set is the 10k set tiled 10x, so every edge (i, j) must satisfy the boundary
    on the 10k matrix entry (i mod 10k, j mod 10k), the edge count must be what that matrix
    predicts, and two half-bands must concatenate to the band."""
    import torch
    n10, n = 10000, 100000
    db10 = engine.SketchDB(big, 16, 14, device=0)
    d10, _ = engine.dist(db10, None, KMERS, tbl1)
    d10 = d10.cpu().numpy()
    x_max, y_max = synth.boundary_for_quantile(d10[::97], 0.02)
    within10 = oracle.assign_threshold(d10, 2, x_max, y_max, threads=8) <= 0        # condensed 10k
    sq = np.zeros((n10, n10), dtype=bool)
    iu = np.triu_indices(n10, 1)
    sq[iu] = within10
    sq |= sq.T
    np.fill_diagonal(sq, True)                      # identical genomes (same index mod 10k): (0,0)
    db = engine.SketchDB(np.tile(big, (10, 1, 1)), 16, 14, device=0)
    b = engine.band_split(n, 0, 8)
    qb, qe = b[5], b[6]
    e, nf = engine.dist_edges(db, None, KMERS, tbl1, slope=2, x_max=x_max, y_max=y_max,
                              q_begin=qb, q_end=qe)
    e = e.cpu().numpy()
    assert int(nf.item()) == 0 and len(e) > 100000
    assert np.all(e[:, 0] >= qb) and np.all(e[:, 0] < qe) and np.all(e[:, 1] > e[:, 0]) and np.all(e[:, 1] < n)
    assert np.all(sq[e[:, 0] % n10, e[:, 1] % n10])                       # every edge is a true edge
    # expected count: for each query row q, refs r in (q, n) with sq[q%10k, r%10k]
    rowsum = sq.sum(axis=1).astype(np.int64)
    csum = np.concatenate([np.zeros((n10, 1), dtype=np.int64), np.cumsum(sq, axis=1)], axis=1)
    want = 0
    for q in range(qb, qe):
        full_blocks_after = (n - 1 - q) // n10                            # whole 10k periods after q
        rem_end = (q + 1) % n10 + (n - 1 - q) % n10                       # remainder window (may wrap)
        s = (q + 1) % n10
        if rem_end <= n10:
            part = csum[q % n10, rem_end] - csum[q % n10, s]
        else:
            part = (csum[q % n10, n10] - csum[q % n10, s]) + csum[q % n10, rem_end - n10]
        want += full_blocks_after * rowsum[q % n10] + part
    assert len(e) == want
    order_ok = np.all((e[1:, 0] > e[:-1, 0]) | ((e[1:, 0] == e[:-1, 0]) & (e[1:, 1] > e[:-1, 1])))
    assert order_ok                                                       # reference row order
    mid = (qb + qe) // 2 // 64 * 64
    e1, _ = engine.dist_edges(db, None, KMERS, tbl1, slope=2, x_max=x_max, y_max=y_max, q_begin=qb, q_end=mid)
    e2, _ = engine.dist_edges(db, None, KMERS, tbl1, slope=2, x_max=x_max, y_max=y_max, q_begin=mid, q_end=qe)
    assert np.array_equal(np.concatenate([e1.cpu().numpy(), e2.cpu().numpy()]), e)
    db.close()
    db10.close()


@pytest.mark.parametrize("s64,kstep", [(16, 1), (156, 1)])
def test_many_kmer_lengths_counts_fallback(s64, kstep):
    """--k-step 1 (17 k-mer lengths): 17 x 11 (or 14) count bits > 128, so the packed per-pair
    state does not fit the count register: the wide-k tile kernel (tests/test_gpu_wide.py) -- until round 5 a
    generic counts -> regression path; same answers."""
    kmers = np.arange(13, 30, kstep, dtype=np.int32)
    n = 150 if s64 == 16 else 40
    sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=14, cluster_size=10, seed=6)
    rng = np.random.Generator(np.random.PCG64(2))
    tbl = (synth.random_match_table(kmers, n_clu=2) * rng.uniform(0.5, 2.0, size=(len(kmers), 2, 2))).astype(np.float32)
    clu = (member % 2).astype(np.uint16)
    got, gf = pp_sketchlib.query_arrays(sk, None, kmers, s64, 14, tbl, clu)
    want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, clu, clu, threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL
    ref, qry = sk[:n - 30], sk[n - 30:]
    got, gf = pp_sketchlib.query_arrays(ref, qry, kmers, s64, 14, tbl, clu[:n - 30], clu[n - 30:])
    want, wf = oracle.query(ref, qry, kmers, s64, 14, tbl, clu[:n - 30], clu[n - 30:], threads=4)
    assert gf == wf and np.abs(got - want).max() <= TOL
    # fused-API edge list of the whole matrix falls back to distances + row-linear compaction
    db = engine.SketchDB(sk, s64, 14, clusters=clu)
    d, _ = engine.dist(db, None, kmers, tbl)
    dn = d.cpu().numpy()
    x_max, y_max = synth.boundary_for_quantile(dn, 0.2)
    e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
    assert np.array_equal(e.cpu().numpy(), oracle.edge_threshold(dn, 2, x_max, y_max))
    # ... and of a band (round 5: the wide-k tile kernel; until then a refusal)
    e1, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, q_begin=0, q_end=n // 2)
    e2, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, q_begin=n // 2, q_end=n)
    assert np.array_equal(np.concatenate([e1.cpu().numpy(), e2.cpu().numpy()]), e.cpu().numpy())
    db.close()


def test_small_job_k_split_is_bit_identical_to_the_tile_epilogue(ppk_option):
    """Below ~350 pair tiles kernel 1 runs one workgroup per (tile, k) and a separate regression
    pass (DESIGN.md 3.1 'Small jobs'); every pair must come out bit for bit as from the fused tile
    epilogue (same expressions in the same order: a result never depends on the job's shape)."""
    import torch
    from poppunk_amd import engine
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl3 = (np.random.Generator(np.random.PCG64(9)).random((5, 3, 3)) * 0.04).astype(np.float32)
    sk, member = synth.make_sketches(900, kmers, cluster_size=40, seed=123, related=False)  # failing fits too
    clu = (member % 3).astype(np.uint16)
    for table, clusters in ((synth.random_match_table(kmers), None), (tbl3, clu)):
        db = engine.SketchDB(sk[:700], 16, 14, clusters=None if clusters is None else clusters[:700])
        dq = engine.SketchDB(sk[700:], 16, 14, clusters=None if clusters is None else clusters[700:])
        res = {}
        for mode in ("0", "100000"):
            ppk_option("ksplit", int(mode))
            a, fa = engine.dist(db, None, kmers, table)
            b, fb = engine.dist(db, dq, kmers, table, q_begin=3, q_end=150)
            res[mode] = (a.clone(), fa, b.clone(), fb)
        ppk_option("ksplit", 640)
        assert res["0"][1] == res["100000"][1] and res["0"][3] == res["100000"][3]
        assert torch.equal(res["0"][0], res["100000"][0]) and torch.equal(res["0"][2], res["100000"][2])
        want, wf = oracle.query(sk[:700], None, kmers, 16, 14, table,
                                ref_clu=None if clusters is None else clusters[:700], threads=4)
        assert res["0"][1] == wf and np.abs(res["0"][0].cpu().numpy() - want).max() <= 1e-6
        db.close()
        dq.close()


@pytest.mark.parametrize("s64,nk", [(16, 5), (16, 3), (16, 6), (156, 5), (40, 2), (8, 11), (2, 9), (1, 5), (3, 4)])
def test_small_jobs_in_one_launch_equal_the_two_pass_path_and_the_tile_kernel(ppk_option, s64, nk):
    """Round 4: a job of less than a round of tiles is ONE launch -- every tile is compared by nk * slices
    workgroups, each leaves its partial counts in scratch and takes a ticket, the last one rebuilds the count
    registers and runs the tile kernel's own epilogue (DESIGN.md 3.1 'Small jobs').  Against the former counts
    pass + regression pass (`ksplit_fused` 0) and the tile kernel itself (`ksplit` 0): bit for bit, for every
    way of cutting a k into pieces, self and ref x query (a few queries against many refs: poppunk_assign),
    bands, several clusters, identical and unrelated samples (counts of nbins and of 0), ragged edges (strip
    tiles), the default sketch size (three-dword count registers); and against the oracle."""
    import torch
    kmers = np.arange(13, 13 + 3 * nk, 3, dtype=np.int32)
    rng = np.random.Generator(np.random.PCG64(s64 * 100 + nk))
    n = 1000 if s64 == 16 else 420
    sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, cluster_size=35, seed=nk + s64, related=(nk != 3))
    sk[7] = sk[3]                                   # duplicates: every bin of every k equal
    sk[n - 1] = sk[n - 2]
    clu = (member % 3).astype(np.uint16)
    tbl3 = (rng.random((nk, 3, 3)) * 0.03).astype(np.float32)
    n_q = 37
    for table, clusters in ((synth.random_match_table(kmers), None), (tbl3, clu)):
        db = engine.SketchDB(sk[:n - n_q], s64, 14, clusters=None if clusters is None else clusters[:n - n_q])
        dq = engine.SketchDB(sk[n - n_q:], s64, 14, clusters=None if clusters is None else clusters[n - n_q:])
        jobs = {"self": lambda: engine.dist(db, None, kmers, table),
                "band": lambda: engine.dist(db, None, kmers, table, q_begin=130, q_end=351),
                "assign": lambda: engine.dist(db, dq, kmers, table),
                "one query": lambda: engine.dist(db, dq, kmers, table, q_begin=5, q_end=6)}
        ppk_option("ksplit", 0)
        want = {k: (a.clone(), int(f)) for k, (a, f) in ((k, fn()) for k, fn in jobs.items())}
        ppk_option("ksplit", 100000)
        ppk_option("ksplit_fused", 0)
        for k, fn in jobs.items():
            a, f = fn()
            assert torch.equal(a, want[k][0]) and int(f) == want[k][1], ("two-pass", k)
        ppk_option("ksplit_fused", 1)
        # ks_grid_pad 1: an odd grid width, which puts the units of a tile on DIFFERENT XCDs (the default width is a
        # multiple of 8: all of a tile's units on one) -- the case the hand-over's agent-scope atomics are there for
        for pad, slices in ((p_, s_) for p_ in (0, 1) for s_ in (0, 1, 2, 4)):
            if slices and s64 % slices:
                continue
            ppk_option("ks_grid_pad", pad)
            ppk_option("ksplit_slices", slices)
            for rep in range(3 if pad == 0 else 2):     # the tile counters must come back to zero after every launch
                for k, fn in jobs.items():
                    a, f = fn()
                    # (a unit needs two blocks: sketches of one block, or of two cut in two, keep the two-pass path)
                    nm = _lib.lib().ppk_last_kernel_name()       # (more than 64 count bits: fitted from the units' parts)
                    assert (nm.endswith(b"k-split fused>") or nm.endswith(b"fit from parts>")) == (s64 >= 2), k
                    assert torch.equal(a, want[k][0]) and int(f) == want[k][1], ("fused", pad, slices, rep, k)
        ppk_option("ks_grid_pad", 0)
        ppk_option("ksplit_slices", 0)
        ref_want, ref_failed = oracle.query(sk[:n - n_q], None, kmers, s64, 14, table,
                                            ref_clu=None if clusters is None else clusters[:n - n_q], threads=8)
        assert want["self"][1] == ref_failed and np.abs(want["self"][0].cpu().numpy() - ref_want).max() <= TOL
        db.close()
        dq.close()


def test_host_call_chunks_the_result_through_bounded_device_memory(ppk_option):
    """ppk_query computes its band in sub-bands through two alternating device buffers (the
    reference CUDA path's device-memory chunking): many tiny sub-bands, one or several devices in
    the list, self and ref x query, all output modes -- identical to the one-piece result."""
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk = synth.make_sketches(1300, kmers, cluster_size=50, seed=31)[0]
    ref, qry = sk[:900], sk[900:]
    base = {}
    for name, (r, q) in (("self", (sk, None)), ("rq", (ref, qry))):
        base[name] = (pp_sketchlib.query_arrays(r, q, kmers, 16, 14, tbl),
                      pp_sketchlib.query_arrays(r, q, kmers, 16, 14, tbl, jaccard=True)[0],
                      pp_sketchlib.query_arrays(r, q, kmers, 16, 14, counts=True)[0])
    ppk_option("chunk_rows", 30000)
    for devices in ((0,), (0, 0, 0)):
        for name, (r, q) in (("self", (sk, None)), ("rq", (ref, qry))):
            d, f = pp_sketchlib.query_arrays(r, q, kmers, 16, 14, tbl, devices=devices)
            assert f == base[name][0][1] and np.array_equal(d, base[name][0][0])
            assert np.array_equal(pp_sketchlib.query_arrays(r, q, kmers, 16, 14, tbl, jaccard=True,
                                                            devices=devices)[0], base[name][1])
            assert np.array_equal(pp_sketchlib.query_arrays(r, q, kmers, 16, 14, counts=True,
                                                            devices=devices)[0], base[name][2])


@pytest.mark.parametrize("s64,nk,words", [(16, 5, 2), (16, 6, 3), (16, 8, 3), (16, 9, 4), (16, 11, 4),
                                          (156, 4, 2), (156, 6, 3), (156, 7, 4), (156, 9, 4)])
def test_count_register_widths(ppk_option, s64, nk, words):
    """The tile kernel keeps each pair's counts in a shift register of 2, 3 or 4 dwords (count k at
    bit (nk-1-k)*bits; bits = 11 at s = 1024, 14 at s = 9984): every width, fields straddling dword
    boundaries, the exact 128-bit limit (9 x 14 = 126), distances and the fused boundary; failed and
    truncated fits (unrelated clusters) read the fields through the general path."""
    bits = 11 if s64 == 16 else 14
    assert -(-nk * bits // 32) == words
    kmers = np.round(np.linspace(13, 31, nk)).astype(np.int32)
    assert len(set(kmers.tolist())) == nk
    n = 330 if s64 == 16 else 290        # > 256: diagonal half tiles and a strip / second ref tile
    for related in (True, False):
        sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=14, cluster_size=30, seed=40 + nk,
                                         related=related)
        tbl = synth.random_match_table(kmers)
        counts, _ = pp_sketchlib.query_arrays(sk, None, kmers, s64, 14, counts=True)
        assert np.array_equal(counts, oracle.match_counts(sk, None, s64, 14, threads=4))
        want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, threads=4)
        ppk_option("ksplit", 0)       # a job this small would take the k-split path
        got, gf = pp_sketchlib.query_arrays(sk, None, kmers, s64, 14, tbl)
        assert gf == wf and np.abs(got - want).max() <= TOL
        ppk_option("ksplit", 640)
        got2, gf2 = pp_sketchlib.query_arrays(sk, None, kmers, s64, 14, tbl)
        assert gf2 == wf and np.array_equal(got2, got)      # k-split: same bits
        if not related:
            assert wf > 0
        db = engine.SketchDB(sk, s64, 14)
        x_max, y_max = synth.boundary_for_quantile(got, 0.2)
        e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
        assert np.array_equal(e.cpu().numpy(), oracle.edge_threshold(got, 2, x_max, y_max))
        db.close()


def test_host_result_pages_are_touched_ahead_of_the_download(ppk_option):
    """ppk_query's helper threads write the first byte of every page of the (fresh) result array
    before the chunked download reaches it: any thread count, arrays that do not start on a page
    boundary, several sub-bands and devices in the list -- the result is the one without them."""
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk = synth.make_sketches(1700, kmers, cluster_size=40, seed=77)[0]      # 1.44 M pairs: 11.6 MB of float2
    ppk_option("prefault_threads", 0)
    want, wf = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl)
    assert want.nbytes > (8 << 20)
    for threads, chunk_rows, devices in (("1", None, (0,)), ("3", "200000", (0,)), ("64", "77777", (0, 0)), ("8", None, (0,))):
        ppk_option("prefault_threads", int(threads))
        if chunk_rows:
            ppk_option("chunk_rows", int(chunk_rows))
        else:
            ppk_option("chunk_rows", 8 << 20)
        got, gf = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl, devices=devices)
        assert gf == wf and np.array_equal(got, want)
    # the square / long helpers pre-touch their results the same way
    ppk_option("prefault_threads", 5)
    n = 2100                                                                 # 17.6 MB square
    v = np.random.Generator(np.random.PCG64(3)).random(n * (n - 1) // 2, dtype=np.float32)
    sq = pp_sketchlib.longToSquare(v.reshape(-1, 1))
    assert sq.shape == (n, n) and np.array_equal(sq[np.triu_indices(n, 1)], v) and np.array_equal(sq, sq.T)
    assert np.array_equal(pp_sketchlib.squareToLong(sq).ravel(), v)


def test_queryDatabase_surface_on_a_reference_layout_h5(tmp_path):
    """The surface PopPUNK calls (PopPUNK/sketchlib.py:475-632) end to end on `.h5` databases in the
    reference's own layout, written and read natively (h5lite): self and ref x query, the multi-cluster
    random table from the ref database's /random group ([EXT] layout) with the queries' clusters taken
    from the nearest centroid, the --plot-fit leg, the resident-database cache across calls, and the
    refusal of a database without random match chances."""
    from poppunk_amd import h5lite, sketchdb, sketchlib
    if not h5lite.available():
        pytest.skip("libhdf5 not found")
    kmers = np.asarray([13, 17, 21, 25, 29], dtype=np.int32)
    sk, member = synth.make_sketches(260, kmers, cluster_size=20, seed=21)
    rn = ["ref_%03d" % i for i in range(200)]
    qn = ["qry_%03d" % i for i in range(60)]
    tbl = np.ascontiguousarray((np.random.Generator(np.random.PCG64(2)).random((5, 2, 2)) * 0.03 *
                                np.asarray([1.0, 0.1, 0.01, 0.001, 0.0001])[:, None, None]).astype(np.float32))
    tbl = (tbl + tbl.transpose(0, 2, 1)) / 2
    rclu = (member[:200] % 2).astype(np.uint16)
    cent = np.asarray([[0.3, 0.2, 0.2, 0.3], [0.2, 0.3, 0.3, 0.2]])
    rfreq = cent[rclu]
    qclu = np.asarray([i % 2 for i in range(60)], dtype=np.uint16)
    qfreq = cent[qclu] + 0.01
    raw = sketchdb.random_to_raw(tbl, rclu, rn, kmers)
    raw["centroids"] = cent
    rp, qp = str(tmp_path / "refdb"), str(tmp_path / "qrydb")
    sketchdb.save_h5(rp + "/refdb", rn, kmers, sk[:200], 16, 14, random_raw=raw, base_freq=rfreq)
    sketchdb.save_h5(qp + "/qrydb", qn, kmers, sk[200:], 16, 14, base_freq=qfreq)       # queries: no /random
    pp_sketchlib.clear_cache()
    # self
    d = sketchlib.queryDatabase(rn, rn, rp, rp, kmers, self=True, number_plot_fits=2)
    want, wf = oracle.query(sk[:200], None, kmers, 16, 14, tbl, rclu, rclu, threads=4)
    assert d.dtype == np.float32 and d.shape == (19900, 2) and np.abs(d - want).max() <= TOL
    for i in (1, 2):
        lines = open(rp + "/refdb_fit_example_%d.tsv" % i).read().strip().split("\n")
        assert lines[2] == "k\traw_jaccard\tcorrected_jaccard" and len(lines) == 3 + 5
        vals = np.asarray([[float(x) for x in ln.split("\t")] for ln in lines[3:]])
        assert np.all(vals[:, 1] >= vals[:, 2]) and np.all(np.diff(vals[:, 1]) < 0.2)
    # ref x query: the queries' clusters come from the ref table's centroids
    d2 = sketchlib.queryDatabase(rn, qn, rp, qp, kmers, self=False, number_plot_fits=1)
    want2, _ = oracle.query(sk[:200], sk[200:], kmers, 16, 14, tbl, rclu, qclu, threads=4)
    assert np.abs(d2 - want2).max() <= TOL
    assert os.path.exists(str(tmp_path / "qrydb_fit_example_1.tsv"))
    # a second call finds the database loaded and resident (same answer, no re-read)
    os.rename(rp + "/refdb.h5", rp + "/refdb.h5.moved")
    try:
        with pytest.raises(RuntimeError):
            sketchlib.queryDatabase(rn[:50], rn[:50], rp, rp, kmers)          # different names: a real read
    finally:
        os.rename(rp + "/refdb.h5.moved", rp + "/refdb.h5")
    assert np.array_equal(sketchlib.queryDatabase(rn, rn, rp, rp, kmers), d)
    # a self query of the query database: it has no random match chances -> refused, not guessed
    with pytest.raises(RuntimeError, match="no random match chances"):
        sketchlib.queryDatabase(qn, qn, qp, qp, kmers)
    assert pp_sketchlib.queryDatabase(qp + "/qrydb", qp + "/qrydb", qn, qn, kmers, False, False, 1, True, 0).shape == (1770, 2)
    pp_sketchlib.clear_cache()


def test_host_call_honours_ctrl_c_and_reports_progress_on_fd2(ppk_option, tmp_path):
    """the bindings being replaced poll for signals in their long loops and raise, and
    print progress to stderr, which PopPUNK silences with an fd-level redirect.  Here: SIGINT during
    a many-sub-band host call -> KeyboardInterrupt well before the job would have finished, library
    intact afterwards; the meter goes to file descriptor 2 (seen through a dup2 redirect, as
    PopPUNK.utils.stderr_redirected does it), only for jobs of several sub-bands, off with progress=0."""
    import signal
    import threading
    import time
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    tbl = synth.random_match_table(kmers)
    sk = synth.make_sketches(6000, kmers, cluster_size=50, seed=6)[0]
    want, _ = pp_sketchlib.query_arrays(sk[:1500], None, kmers, 16, 14, tbl)
    ppk_option("chunk_rows", 20000)            # 6000 genomes: ~900 sub-bands
    ppk_option("progress", 2)                  # 2: also for jobs below the 2^29-row (~0.1 s) threshold

    def fd2_of(fn):
        path = str(tmp_path / "fd2.txt")
        saved = os.dup(2)
        with open(path, "wb") as f:
            os.dup2(f.fileno(), 2)
            try:
                fn()
            finally:
                os.dup2(saved, 2)
                os.close(saved)
        return open(path, "rb").read()

    text = fd2_of(lambda: pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl))
    assert b"Progress (GPU): " in text and text.rstrip().endswith(b"100.0%")
    ppk_option("progress", 0)
    assert b"Progress" not in fd2_of(lambda: pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl))
    ppk_option("progress", 1)
    assert b"Progress" not in fd2_of(lambda: pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl))     # small job: silent
    ppk_option("chunk_rows", 8 << 20)
    assert b"Progress" not in fd2_of(lambda: pp_sketchlib.query_arrays(sk[:1500], None, kmers, 16, 14, tbl))
    # Ctrl-C (sub-bands of 64 queries: a few thousand of them, so the job lasts long enough to be cut short)
    ppk_option("chunk_rows", 1000)
    ppk_option("progress", 0)
    t0 = time.perf_counter()
    pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl)
    full = time.perf_counter() - t0
    timer = threading.Timer(full * 0.1, lambda: os.kill(os.getpid(), signal.SIGINT))
    timer.start()
    t0 = time.perf_counter()
    with pytest.raises(KeyboardInterrupt):
        pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl)
    took = time.perf_counter() - t0
    timer.join()
    assert took < 0.9 * full, (took, full)
    assert signal.getsignal(signal.SIGINT) is signal.default_int_handler      # the handler was restored
    ppk_option("chunk_rows", 8 << 20)
    again, _ = pp_sketchlib.query_arrays(sk[:1500], None, kmers, 16, 14, tbl)
    assert np.array_equal(again, want)


def test_fit_tables_are_reused_only_for_identical_inputs(sk300):
    """The device keeps the log-J / (E, F) tables of the last call and skips the rebuild when the k
    list, random table, sketch size and [EXT] options are the same: alternate between inputs (and
    through the modes that do not build them) and check every answer."""
    sk = sk300[0]
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    db = engine.SketchDB(sk, 16, 14)
    t1 = synth.random_match_table(kmers)
    t2 = synth.random_match_table(kmers, genome_length=5_000_000)
    assert not np.array_equal(t1, t2)
    want = {}
    for name, tbl, rc in (("t1", t1, True), ("t2", t2, True), ("none", None, False)):
        want[name] = oracle.query(sk, None, kmers, 16, 14, tbl, random_correct=rc, threads=4)[0]
    order = ["t1", "t1", "t2", "none", "t1", "counts", "t1", "t2", "t2", "jaccard", "t2", "none", "none"]
    tables = {"t1": (t1, True), "t2": (t2, True), "none": (None, False)}
    for step in order:
        if step == "counts":
            c, _ = engine.dist(db, None, kmers, counts=True)
            assert np.array_equal(c.cpu().numpy(), oracle.match_counts(sk, None, 16, 14, threads=4))
        elif step == "jaccard":
            j, _ = engine.dist(db, None, kmers, t1, jaccard=True)
            assert np.array_equal(j.cpu().numpy(), oracle.query(sk, None, kmers, 16, 14, t1, jaccard=True, threads=4)[0])
        else:
            tbl, rc = tables[step]
            d, _ = engine.dist(db, None, kmers, tbl, random_correct=rc)
            assert np.abs(d.cpu().numpy() - want[step]).max() <= TOL, step
    # a different k list with the same number of k
    k2 = kmers + 1
    d, _ = engine.dist(db, None, k2, synth.random_match_table(k2))
    assert np.abs(d.cpu().numpy() - oracle.query(sk, None, k2, 16, 14, synth.random_match_table(k2), threads=4)[0]).max() <= TOL
    db.close()


def test_interior_tiles_with_identical_and_unrelated_pairs(tbl1, ppk_option):
    """The epilogue of an interior tile (off the diagonal, default sketch shape, one cluster pair) reads
    the (E, F) table from LDS, where only counts 0..1023 have a row: a count of 1024 -- every bin of
    a k equal -- wraps to row 0, the NaN sentinel, and the wavefront takes the general path.  Samples
    copied far away in the index space (exact duplicates, duplicates at some k only) and unrelated
    samples (every fit fails) sit in interior tiles here, next to ordinary pairs; ref x query too."""
    ppk_option("ksplit", 0)          # the tile kernel's own epilogue, not the small-job path
    n = 1500
    sk, _ = synth.make_sketches(n, KMERS, cluster_size=40, seed=99)
    rng = np.random.Generator(np.random.PCG64(5))
    sk = sk.copy()
    sk[700] = sk[10]                                   # all five k identical
    sk[1301, :2] = sk[45, :2]                          # identical at the two smallest k only
    sk[900, 4] = sk[3, 4]                              # identical at the largest k only
    for s in (600, 1100, 1499):                        # unrelated to everything: all fits fail
        sk[s] = rng.integers(0, np.iinfo(np.int64).max, size=sk[s].shape, dtype=np.int64).astype(np.uint64)
    got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, random_tbl=tbl1, threads=8)
    assert gf == wf and wf > 0
    assert np.abs(got - want).max() <= TOL
    row = 10 * n - 10 * 11 // 2 + (700 - 10 - 1)
    assert np.array_equal(got[row], [0.0, 0.0])
    ref, qry = sk[:1024], sk[1024:]
    got, gf = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, tbl1)
    want, wf = oracle.query(ref, qry, KMERS, 16, 14, random_tbl=tbl1, threads=8)
    assert gf == wf
    assert np.abs(got - want).max() <= TOL
    # three random-match clusters in contiguous runs (a tile with ONE cluster pair takes the LDS path with
    # that pair's block of the table; tiles across a run boundary do not)
    tbl3 = (rng.random((5, 3, 3)) * 0.04).astype(np.float32)
    clu = np.zeros(n, dtype=np.uint16)
    clu[500:1100] = 2
    clu[1100:] = 1
    clu[37] = 1                                        # one stray sample inside a run
    got3, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl3, ref_clusters=clu)
    want3, wf = oracle.query(sk, None, KMERS, 16, 14, random_tbl=tbl3, ref_clu=clu, threads=8)
    assert gf == wf
    assert np.abs(got3 - want3).max() <= TOL
    got, gf = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, tbl3, ref_clusters=clu[:1024], qry_clusters=clu[1024:])
    want, wf = oracle.query(ref, qry, KMERS, 16, 14, random_tbl=tbl3, ref_clu=clu[:1024], qry_clu=clu[1024:], threads=8)
    assert gf == wf
    assert np.abs(got - want).max() <= TOL
    # the interior path off (option "lds_table" 0): bit-identical results either way
    a, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1)
    ppk_option("lds_table", 0)
    b, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl1)
    assert np.array_equal(a, b)
    b3, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl3, ref_clusters=clu)
    assert np.array_equal(got3, b3)


@pytest.mark.parametrize("nk", [3, 4])
def test_interior_tiles_with_three_and_four_kmer_lengths(ppk_option, nk):
    """11-bit counts of 3 or 4 k also fill two dwords: the LDS-table epilogue shifts the register up
    to the 5-k layout and the missing k read a (1, 1) row.  Same bits as the general path, oracle
    within tolerance, fused boundary and neighbours included."""
    ppk_option("ksplit", 0)
    kmers = np.asarray([13, 17, 21, 25][:nk], dtype=np.int32)
    n = 1300
    sk, _ = synth.make_sketches(n, kmers, cluster_size=40, seed=70 + nk)
    sk = sk.copy()
    sk[900] = sk[20]                                   # an exact duplicate far away
    tbl = synth.random_match_table(kmers)
    got, gf = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl)
    want, wf = oracle.query(sk, None, kmers, 16, 14, random_tbl=tbl, threads=8)
    assert gf == wf
    assert np.abs(got - want).max() <= TOL
    db = engine.SketchDB(sk, 16, 14)
    x_max, y_max = synth.boundary_for_quantile(got, 0.1)
    e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
    assert np.array_equal(e.cpu().numpy(), oracle.edge_threshold(got, 2, x_max, y_max))
    gi, gj, gd = engine.knn_from_sketches(db, kmers, tbl, 4, method="tiles")
    wi, wj, wd = oracle.knn(oracle.long_to_square(got[:, 0]), 4)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gj.cpu().numpy(), wj)
    assert np.array_equal(gd.cpu().numpy(), wd)
    db.close()
    ppk_option("lds_table", 0)
    b, _ = pp_sketchlib.query_arrays(sk, None, kmers, 16, 14, tbl)
    assert np.array_equal(got, b)


@pytest.mark.parametrize("bbits,s64", [(14, 16), (8, 5)])
def test_bands_larger_than_one_dispatch_go_out_as_several_launches(ppk_option, bbits, s64):
    """A dispatch holds fewer than 2^32 work-items (8.4 M pair tiles of 512 threads): 370 000 genomes against
    themselves exceed it.  The launcher then cuts the band into consecutive query-row pieces, each writing to
    its own part of the output.  Forced here with a tiny tile budget: distances, counts, Jaccard, the fused
    edge list, neighbours from the tiles, self and ref x query, all equal to the single-launch results."""
    kmers = np.asarray([13, 17, 21, 25], dtype=np.int32)
    sk, _ = synth.make_sketches(700, kmers, sketchsize64=s64, bbits=bbits, cluster_size=35, seed=77)
    tbl = synth.random_match_table(kmers)
    ref, qry = sk[:450], sk[450:]
    ppk_option("ksplit", 0)
    single = {}
    for pieces in (False, True):
        if pieces:
            ppk_option("launch_tiles", 8)        # 2 ref tiles of 256 -> 4 tiles of 32 query rows -> 128 -> 64-row pieces
        out = {}
        out["d_self"] = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, tbl)
        out["c_self"] = pp_sketchlib.query_arrays(sk, None, kmers, s64, bbits, counts=True)
        out["j_rq"] = pp_sketchlib.query_arrays(ref, qry, kmers, s64, bbits, tbl, jaccard=True)
        out["d_rq"] = pp_sketchlib.query_arrays(ref, qry, kmers, s64, bbits, tbl)
        db = engine.SketchDB(sk, s64, bbits)
        rdb, qdb = engine.SketchDB(ref, s64, bbits), engine.SketchDB(qry, s64, bbits)
        band, _ = engine.dist(db, None, kmers, tbl, q_begin=100, q_end=613)
        out["band"] = (band.cpu().numpy(), 0)
        x_max, y_max = synth.boundary_for_quantile(out["d_self"][0], 0.1)
        e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
        out["e_self"] = (e.cpu().numpy(), 0)
        e, _ = engine.dist_edges(rdb, qdb, kmers, tbl, slope=1, x_max=x_max, y_max=y_max, q_begin=3, q_end=250)
        out["e_rq"] = (e.cpu().numpy(), 0)
        if bbits == 14:
            i, j, d = engine.knn_from_sketches(db, kmers, tbl, 5, method="tiles")
            out["knn"] = (np.stack([i.cpu().numpy(), j.cpu().numpy()]), 0)
            out["knn_d"] = (d.cpu().numpy(), 0)
        for x in (db, rdb, qdb):
            x.close()
        if not pieces:
            single = out
            want, _ = oracle.query(sk, None, kmers, s64, bbits, tbl, threads=4)
            assert np.abs(out["d_self"][0] - want).max() <= 1e-6
        else:
            for key in single:
                assert out[key][1] == single[key][1], key
                assert np.array_equal(out[key][0], single[key][0]), key
            assert len(single["e_self"][0]) > 100 and len(single["e_rq"][0]) > 10


@pytest.mark.parametrize("s64,nk,n", [(16, 5, 1000), (16, 5, 3100), (16, 3, 700), (16, 8, 1300), (156, 5, 600), (40, 6, 900)])
def test_fused_edge_list_through_the_k_split_path_equals_the_tile_kernels(ppk_option, s64, nk, n):
    """Round 5: small fused distance -> boundary -> edge-list jobs run one workgroup per (tile, k) like small distance
    jobs (the tile's last unit applies the boundary: from the LDS table at s = 1 024, from the units' parts
    elsewhere).  Self with diagonal / strip tiles and bands, ref x query, several random-match clusters, both
    predicates: the tile kernel's list, element for element, and the oracle's."""
    import torch
    from poppunk_amd import engine
    kmers = np.round(np.linspace(13, 29, nk)).astype(np.int32)
    sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=14, cluster_size=25, seed=n + nk)
    sk[7] = sk[6]
    clu = np.sort((member % 3).astype(np.uint16))
    rng = np.random.Generator(np.random.PCG64(5))
    tbl = (synth.random_match_table(kmers, n_clu=3) * rng.uniform(0.5, 2.0, size=(nk, 3, 3))).astype(np.float32)
    db = engine.SketchDB(sk, s64, 14, clusters=clu)
    nr = n - n // 5
    rdb, qdb = engine.SketchDB(sk[:nr], s64, 14, clusters=clu[:nr]), engine.SketchDB(sk[nr:], s64, 14, clusters=clu[nr:])
    d, _ = engine.dist(db, None, kmers, tbl)
    dn = d.cpu().numpy()
    x_max, y_max = synth.boundary_for_quantile(dn, 0.1)
    name = lambda: _lib.lib().ppk_last_kernel_name().decode()

    def lists():
        out = []
        for inclusive in (True, False):
            out.append(engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, inclusive=inclusive)[0])
            nm = name()
            out.append(torch.cat([engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, inclusive=inclusive,
                                                    q_begin=a, q_end=b)[0] for a, b in ((0, 33), (33, n // 2 + 1), (n // 2 + 1, n))]))
            out.append(engine.dist_edges(rdb, qdb, kmers, tbl, slope=1, x_max=x_max, y_max=y_max, inclusive=inclusive)[0])
        return nm, out

    nm_ks, ks = lists()
    assert "k-split" in nm_ks, nm_ks
    ppk_option("ksplit", 0)
    nm_tile, tile = lists()
    assert "k-split" not in nm_tile
    for a, b in zip(ks, tile):
        assert torch.equal(a, b)
    assert np.array_equal(tile[0].cpu().numpy(), oracle.edge_threshold(dn, 2, x_max, y_max))
    assert torch.equal(ks[0], ks[1]) and len(ks[0]) > 100
    for x in (db, rdb, qdb):
        x.close()
