"""k-mer lists wider than the tile kernel's 128-bit count register (nk x count bits > 128).

PopPUNK computes any np.arange(min_k, max_k + 1, k_step) (PopPUNK/__main__.py:77-80,299) and its documentation
recommends such lists: k = 6..15 for beta-coronaviruses, k-step "two or three" for accuracy, larger sketches for
close genomes (docs/sketching.rst:123-139,152-156).  At the default sketch size (9 984 bins, 14-bit counts) ten or
more k-mer lengths no longer fit the register; the WIDE instantiation of the tile kernel windows it over the k list
(ppk_dist.hip PackWide).  Everything the register path offers must work there too: distances, the fused on-device
edge list (whole jobs and bands), neighbours from the tiles, the host calls.

Two kinds of check:
  * HIP against the CPU oracle on the documented lists at s = 9 984, at a size with many ref tiles;
  * the wide path FORCED (option "wide_kpg") onto k lists the register holds: the two paths evaluate the same
    expressions in the same order, so every bit must agree.
"""
import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import engine, pp_sketchlib, synth

pytestmark = pytest.mark.gpu

TOL = 1e-6
THREADS = 16

# the lists the documentation names (10 / 10 / 12 / 17 k-mer lengths)
K_CORONA = np.arange(6, 16, 1, dtype=np.int32)          # docs/sketching.rst:123-139: k = 6..15
K_STEP2 = np.arange(13, 32, 2, dtype=np.int32)          # --k-step 2, 13..31
K_12 = np.arange(7, 30, 2, dtype=np.int32)              # 12 lengths
K_STEP1 = np.arange(13, 30, 1, dtype=np.int32)          # --k-step 1, 13..29: 17 lengths


def _table(kmers, n_clu=1, seed=2):
    # (genomes of 20 kb for the lists that start below k = 10 -- the coronavirus row of docs/sketching.rst:123-139; with
    # 2 Mb genomes every 6-mer matches by chance, J_r = 1 exactly: that case has its own test below)
    tbl = synth.random_match_table(kmers, genome_length=20_000 if min(kmers) < 10 else 2_000_000, n_clu=n_clu)
    if n_clu > 1:
        rng = np.random.Generator(np.random.PCG64(seed))
        tbl = (tbl * rng.uniform(0.5, 2.0, size=tbl.shape)).astype(np.float32)
    return tbl


def _close(got, want, what):
    diff = np.abs(got - want)
    worst = float(diff.max()) if diff.size else 0.0
    n_diff = int(np.count_nonzero(diff))
    print("%s: %d rows, max |d - oracle| = %.3g, %d values differ" % (what, len(got), worst, n_diff))
    assert worst <= TOL, what
    assert n_diff <= max(20, len(got) // 5000), (what, n_diff)


@pytest.mark.parametrize("kmers", [K_CORONA, K_STEP2, K_12, K_STEP1], ids=["k6-15", "k13-31s2", "k7-29s2", "k13-29s1"])
def test_documented_k_lists_at_the_default_sketch_size(kmers, ppk_option):
    """2 700 genomes (11 ref tiles, a ragged right edge, diagonal half tiles) at s = 9 984: distances (through the wide
    tile kernel and through the k-split path long sketches take by default: same bits), the fused edge list of the
    whole job and of three bands, neighbours from the tiles -- against the oracle."""
    n, s64 = 2700, 156
    sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=14, cluster_size=30, seed=len(kmers))
    clu = (member % 2).astype(np.uint16)
    tbl = _table(kmers, n_clu=2)
    want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, clu, clu, threads=THREADS)
    db = engine.SketchDB(sk, s64, 14, clusters=clu)
    d_ks, f_ks = engine.dist(db, None, kmers, tbl)          # default: one workgroup per (tile, k) + the fit pass
    assert not engine._lib.lib().ppk_last_kernel_name().decode().endswith("wide>")
    ppk_option("ksplit", 0)
    d, f = engine.dist(db, None, kmers, tbl)
    assert engine._lib.lib().ppk_last_kernel_name().decode().endswith("wide>")
    ppk_option("ksplit", 1200)
    assert int(f.item()) == wf == int(f_ks.item())
    assert np.array_equal(d.cpu().numpy().view(np.uint32), d_ks.cpu().numpy().view(np.uint32))
    _close(d.cpu().numpy(), want, "distances nk=%d" % len(kmers))
    # fused edges: whole job, then bands (none of them aligned to the 32-query tile)
    got = d.cpu().numpy()
    x_max, y_max = synth.boundary_for_quantile(got, 0.05)
    for inclusive in (True, False):
        ref_edges = oracle.edge_threshold(got, 2, x_max, y_max, inclusive=inclusive)
        e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, inclusive=inclusive)
        assert np.array_equal(e.cpu().numpy(), ref_edges)
        parts = []
        for qb, qe in ((0, 777), (777, 2001), (2001, n)):
            eb, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, inclusive=inclusive,
                                      q_begin=qb, q_end=qe)
            parts.append(eb.cpu().numpy())
        assert np.array_equal(np.concatenate(parts), ref_edges)
    # the oracle's own distances give the same list except where a distance differs in its last bit AND sits on
    # the boundary: none here
    assert np.array_equal(oracle.edge_threshold(want, 2, x_max, y_max), oracle.edge_threshold(got, 2, x_max, y_max))
    # neighbours from the tiles
    for knn, col in ((5, 0), (12, 1)):
        wi, wj, wd = oracle.knn(oracle.long_to_square(want[:, col]), knn)
        gi, gj, gd = (x.cpu().numpy() for x in engine.knn_from_sketches(db, kmers, tbl, knn, dist_col=col, method="tiles"))
        assert np.array_equal(gi, wi) and np.abs(gd - wd).max() <= TOL
        same = gd == wd
        assert same.mean() > 0.999 and np.array_equal(gj[same], wj[same])
    db.close()


def test_wide_ref_query_and_host_calls():
    """ref x query (what poppunk_assign runs) with a wide list: device call, host call (ppk_query), the fused host
    edge call with its band pieces, the row order q * n_ref + r."""
    kmers, s64 = K_CORONA, 156
    sk, member = synth.make_sketches(1500, kmers, sketchsize64=s64, bbits=14, cluster_size=25, seed=11)
    ref, qry = sk[:1100], sk[1100:]
    tbl = _table(kmers)
    want, wf = oracle.query(ref, qry, kmers, s64, 14, tbl, threads=THREADS)
    got, gf = pp_sketchlib.query_arrays(ref, qry, kmers, s64, 14, tbl)
    assert gf == wf
    _close(got, want, "ref x query nk=10")
    x_max, y_max = synth.boundary_for_quantile(got, 0.05)
    ref_edges = oracle.edge_threshold(got, 2, x_max, y_max, n_ref=1100)
    rdb, qdb = engine.SketchDB(ref, s64, 14), engine.SketchDB(qry, s64, 14)
    e, _ = engine.dist_edges(rdb, qdb, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
    assert np.array_equal(e.cpu().numpy(), ref_edges)
    he, hf = pp_sketchlib.query_edges_arrays(ref, qry, kmers, s64, 14, 2, x_max, y_max, random_table=tbl)
    assert np.array_equal(he, ref_edges) and hf == wf
    # neighbours, ref x query
    rect = want[:, 0].reshape(400, 1100)
    gi, gj, gd = (x.cpu().numpy() for x in engine.knn_ref_query(rdb, qdb, kmers, tbl, 3))
    gj = gj.reshape(1500, 3)
    for r in (0, 500, 1099):
        assert gj[r].tolist() == (np.argsort(rect[:, r], kind="stable")[:3] + 1100).tolist()
    for q in (0, 399):
        assert gj[1100 + q].tolist() == np.argsort(rect[q], kind="stable")[:3].tolist()
    rdb.close()
    qdb.close()


@pytest.mark.parametrize("s64,kmers,kpg", [(16, synth.DEFAULT_KMERS, 1), (16, synth.DEFAULT_KMERS, 2), (16, (13, 16, 19, 22, 25, 28, 31), 3),
                                            (16, tuple(range(11, 32)), 0), (156, synth.DEFAULT_KMERS, 2), (156, tuple(range(12, 30, 2)), 4)])
def test_forced_wide_path_returns_the_register_paths_bits(ppk_option, s64, kmers, kpg):
    """Option "wide_kpg" narrows the window, sending k lists the register holds through the wide path: distances,
    edge lists (self with strip tiles and bands; ref x query) and neighbour lists are bit-identical.
    (kpg 0 with 21 lengths of 11-bit counts: naturally wide at s = 1 024, compared with the two-pass counts
    route's arithmetic through the oracle instead.)"""
    kmers = np.asarray(kmers, dtype=np.int32)
    n = 1300 if s64 == 16 else 600
    sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=14, cluster_size=20, seed=s64 + kpg)
    sk[5] = sk[4]                                   # identical pair: every count = nbins
    clu = (member % 3).astype(np.uint16)
    tbl = _table(kmers, n_clu=3)
    db = engine.SketchDB(sk, s64, 14, clusters=clu)
    qdb = engine.SketchDB(sk[:300], s64, 14, clusters=clu[:300])
    ppk_option("ksplit", 0)                          # (the small-job path has its own tests)

    def run():
        d, f = engine.dist(db, None, kmers, tbl)
        name = engine._lib.lib().ppk_last_kernel_name().decode()
        dn = d.cpu().numpy()
        x_max, y_max = synth.boundary_for_quantile(dn, 0.1)
        e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
        eb, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, q_begin=100, q_end=n - 77)
        rq, _ = engine.dist(db, qdb, kmers, tbl, q_begin=3, q_end=290)
        erq, _ = engine.dist_edges(db, qdb, kmers, tbl, slope=1, x_max=x_max, y_max=y_max, inclusive=False)
        knn = [x.cpu().numpy() for x in engine.knn_from_sketches(db, kmers, tbl, 6, dist_col=0, method="tiles")]
        return name, dn, int(f.item()), e.cpu().numpy(), eb.cpu().numpy(), rq.cpu().numpy(), erq.cpu().numpy(), knn

    if kpg:
        base = run()
        assert not base[0].endswith("wide>")
        ppk_option("wide_kpg", kpg)
        wide = run()
        assert wide[0].endswith("wide>")
        assert np.array_equal(base[1].view(np.uint32), wide[1].view(np.uint32)) and base[2] == wide[2]
        for a, b in zip(base[3:7], wide[3:7]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
        for a, b in zip(base[7], wide[7]):
            assert np.array_equal(a, b)
    else:
        wide = run()
        assert wide[0].endswith("wide>")
        want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, clu, clu, threads=THREADS)
        assert wide[2] == wf
        _close(wide[1], want, "21 lengths at s = 1 024")
    db.close()
    qdb.close()


def test_random_match_chance_of_exactly_one_gives_zero_distances_not_failed_fits():
    """k = 6 on 2 Mb genomes: J_r = 1.0 in float32, observed_excess is 0 / 0.  Upstream's `jaccard < tolerance` is false
    for a NaN, so the point stays in the fit, the slope is NaN and both distances are 0 -- NOT a failed fit (the oracle
    restates exactly that).  Wide path and register path alike."""
    for kmers, s64 in ((K_CORONA, 156), (np.asarray([6, 13, 17, 21, 25], dtype=np.int32), 16)):
        sk, _ = synth.make_sketches(600, kmers, sketchsize64=s64, bbits=14, cluster_size=30, seed=2)
        tbl = synth.random_match_table(kmers, genome_length=2_000_000)
        assert tbl.ravel()[0] == 1.0
        want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, threads=THREADS)
        got, gf = pp_sketchlib.query_arrays(sk, None, kmers, s64, 14, tbl)
        assert wf == 0 and gf == 0 and not want.any()
        assert np.array_equal(got, want)
        # ref x query, and the small-job path's fit
        got, gf = pp_sketchlib.query_arrays(sk[:500], sk[500:], kmers, s64, 14, tbl)
        assert gf == 0 and not got.any()


def test_wide_unrelated_clusters_failed_fits():
    """Unrelated clusters: almost every pair has J below the floor at the first k -- the general statement of the
    fit reading every group back from the slot -- and the failed-fit count."""
    kmers, s64 = K_12, 156
    sk, _ = synth.make_sketches(700, kmers, sketchsize64=s64, bbits=14, cluster_size=35, seed=9, related=False)
    tbl = _table(kmers)
    want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, threads=THREADS)
    got, gf = pp_sketchlib.query_arrays(sk, None, kmers, s64, 14, tbl)
    assert wf > 200000 and gf == wf
    assert np.abs(got - want).max() <= TOL


def test_wide_slots_are_recycled_across_many_more_tiles_than_slots():
    """More pair tiles than spill slots in flight over a launch (9 000 genomes at s = 1 024 with 21 lengths: 5 100
    tiles, 1 024 slots, 512 resident workgroups): the pool's bitmap is back at zero afterwards and a second run
    returns the same bits; sampled rows agree with the oracle."""
    import torch
    kmers = np.arange(11, 32, dtype=np.int32)
    n = 9000
    sk, _ = synth.make_sketches(n, kmers, sketchsize64=16, bbits=14, cluster_size=45, seed=3)
    tbl = _table(kmers)
    db = engine.SketchDB(sk, 16, 14)
    d1, f1 = engine.dist(db, None, kmers, tbl)
    assert engine._lib.lib().ppk_last_kernel_name().decode().endswith("wide>")      # (s = 1 024: 5 100 tiles are no small job)
    a = d1.cpu().numpy().copy()
    d2, f2 = engine.dist(db, None, kmers, tbl, out=d1)
    torch.cuda.synchronize()
    assert np.array_equal(a.view(np.uint32), d2.cpu().numpy().view(np.uint32)) and int(f1.item()) == int(f2.item())
    # rows of 40 queries spread over the job against the oracle
    qs = np.linspace(0, n - 2, 40).astype(np.int64)
    for q in qs:
        want, _ = oracle.query(sk[q + 1:], sk[q:q + 1], kmers, 16, 14, tbl, threads=THREADS)
        row0 = q * n - q * (q + 1) // 2
        assert np.abs(a[row0:row0 + n - q - 1] - want).max() <= TOL
    db.close()


def test_one_100000_genome_band_with_ten_kmer_lengths():
    """BASELINE config 5's shape with a documented wide list (k = 6..15) at the default sketch size: one band of
    query rows of the 100 000-genome self job through the fused edge kernel, edge for edge against the oracle
    (17.5 GB of sketches: the band is what one of 8 GPUs' sub-bands looks like)."""
    import torch
    n, s64, kmers = 100000, 156, K_CORONA
    t = synth.make_sketches_device(n, kmers, sketchsize64=s64, seed=50, device="cuda:0", chunk=512)
    db = engine.SketchDB(t, s64, 14)
    tbl = _table(kmers)
    qb, qe = 61440, 61504
    # host copies: the band's queries, and every ref (the oracle needs them all)
    sk = np.empty((n, len(kmers), s64 * 14), dtype=np.uint64)
    for s in range(0, n, 10000):
        sk[s:s + 10000] = t[s:s + 10000].cpu().numpy().view(np.uint64)
    del t
    torch.cuda.empty_cache()
    sub, _ = oracle.query(sk[:600], None, kmers, s64, 14, tbl, threads=THREADS)
    x_max, y_max = synth.boundary_for_quantile(sub, 0.02)
    rect, wf = oracle.query(sk, sk[qb:qe], kmers, s64, 14, tbl, threads=THREADS)     # row = (q - qb) * n + r
    a = oracle.assign_threshold(rect, 2, x_max, y_max, threads=THREADS).reshape(qe - qb, n)
    d, f = engine.dist(db, None, kmers, tbl, q_begin=qb, q_end=qe)          # (a band of ~400 pair tiles: the k-split path's two-pass form)
    _lib_name = engine._lib.lib().ppk_last_kernel_name().decode()
    engine._lib.set_option("ksplit", 0)
    try:
        d2, _ = engine.dist(db, None, kmers, tbl, q_begin=qb, q_end=qe)
        assert engine._lib.lib().ppk_last_kernel_name().decode().endswith("wide>") and not _lib_name.endswith("wide>")
    finally:
        engine._lib.set_option("ksplit", 1200)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), d2.cpu().numpy().view(np.uint32))
    got = d.cpu().numpy()
    rows = np.concatenate([rect.reshape(qe - qb, n, 2)[q - qb, q + 1:] for q in range(qb, qe)])
    _close(got, rows, "100 000-genome band, nk = 10")
    for inclusive in (True, False):
        e, nf = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max, inclusive=inclusive,
                                  q_begin=qb, q_end=qe, cap=1 << 20)
        qq, rr = np.nonzero((a <= 0) if inclusive else (a < 0))
        sel = rr > qq + qb
        want = np.stack([qq[sel] + qb, rr[sel]], axis=1).astype(np.int64)
        assert len(want) > 1000
        assert np.array_equal(e.cpu().numpy(), want), (inclusive, len(e), len(want))
    db.close()


def test_small_wide_jobs_take_the_k_split_path_and_return_the_tile_kernels_bits(ppk_option):
    """poppunk_assign with a handful of genomes at the default sketch size and a wide k list: a job of less than a
    round of pair tiles runs one workgroup per (tile, k) and a fit pass (two launches: the counts of a wide list never
    enter a register) -- same expressions in the same order as the tile kernel's epilogue, so the same bits."""
    kmers, s64 = K_CORONA, 156
    sk, member = synth.make_sketches(1400, kmers, sketchsize64=s64, bbits=14, cluster_size=20, seed=21)
    clu = (member % 2).astype(np.uint16)
    tbl = _table(kmers, n_clu=2)
    rdb = engine.SketchDB(sk[:1000], s64, 14, clusters=clu[:1000])
    name = lambda: engine._lib.lib().ppk_last_kernel_name().decode()
    for q in (1, 7, 64, 400):
        qdb = engine.SketchDB(sk[1000:1000 + q], s64, 14, clusters=clu[1000:1000 + q])
        a, fa = engine.dist(rdb, qdb, kmers, tbl)
        small = not name().endswith("wide>")
        ppk_option("ksplit", 0)
        b, fb = engine.dist(rdb, qdb, kmers, tbl)
        assert name().endswith("wide>")
        ppk_option("ksplit", 1200)
        assert small, q                                    # (4 ref tiles x ceil(q / 32) query tiles against 215 * 5 / 10 = 107)
        assert torch_equal_bits(a, b) and int(fa.item()) == int(fb.item())
        want, wf = oracle.query(sk[:1000], sk[1000:1000 + q], kmers, s64, 14, tbl, clu[:1000], clu[1000:1000 + q], threads=THREADS)
        assert int(fa.item()) == wf and np.abs(a.cpu().numpy() - want).max() <= TOL
        qdb.close()
    # self, with every way of cutting a k into pieces
    sdb = engine.SketchDB(sk[:500], s64, 14, clusters=clu[:500])
    ppk_option("ksplit", 0)
    base, _ = engine.dist(sdb, None, kmers, tbl)
    ppk_option("ksplit", 1200)
    for slices in (0, 1, 2, 4):
        ppk_option("ksplit_slices", slices)
        got, _ = engine.dist(sdb, None, kmers, tbl)
        assert name().endswith("fit from parts>")
        assert torch_equal_bits(got, base), slices
    # ... long sketches take that path at any size its scratch allows (option "ksplit_long"), other shapes up to 700
    # tiles (1 400 self: 176 tiles); "ksplit" 0 is the tile kernel
    big = engine.SketchDB(sk, s64, 14, clusters=clu)
    engine.dist(big, None, kmers, tbl)
    assert not name().endswith("wide>")
    ppk_option("ksplit_long", 0)
    engine.dist(big, None, kmers, tbl)
    assert not name().endswith("wide>")
    ppk_option("ksplit", 0)
    engine.dist(big, None, kmers, tbl)
    assert name().endswith("wide>")
    ppk_option("ksplit", 1200)
    for db in (rdb, sdb, big):
        db.close()
    # the same fit-from-parts unit kernel on a list the register holds (option "wide_kpg"): the register path's bits
    k5 = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    for s64_, n_ in ((156, 700), (16, 900)):
        sk5, _ = synth.make_sketches(n_, k5, sketchsize64=s64_, bbits=14, cluster_size=20, seed=5)
        sk5[3] = sk5[2]
        t5 = _table(k5)
        d5 = engine.SketchDB(sk5, s64_, 14)
        ppk_option("wide_kpg", 0)
        a, fa = engine.dist(d5, None, k5, t5)
        assert name().endswith("k-split fused>") == (s64_ == 16)      # (70 count bits at s = 9 984: from the parts anyway)
        ppk_option("wide_kpg", 2)
        b, fb = engine.dist(d5, None, k5, t5)
        assert name().endswith("fit from parts>")
        ppk_option("wide_kpg", 0)
        assert torch_equal_bits(a, b) and int(fa.item()) == int(fb.item())
        d5.close()


def torch_equal_bits(a, b):
    import torch
    return bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))


@pytest.mark.parametrize("shape", ["default sketch, 5 k", "default sketch, 10 k (fit from parts)", "s=1024, 5 k"])
def test_ksplit_handover_with_a_tile_s_units_on_different_xcds(ppk_option, shape):
    """The one-launch k-split path hands a tile's partial counts from its units to the unit that draws the last
    ticket through agent-scope atomics (ppk_dist.hip, "Visibility between workgroups").  The product grid is a
    multiple of 8 wide, so under today's dispatch (workgroup -> XCD = linear id mod 8, tools/ubench_grid_xcd.hip) all
    units of a tile share one XCD's L2 and the protocol is never needed.  Option "ks_grid_pad" makes the grid one
    empty column wider: unit y of tile x then runs on XCD (x + y) mod 8.  Distances (MODE_DIST) and the fused edge
    list (MODE_MASK), repeated (the tickets must return to zero), against the tile kernel's bits and the oracle."""
    if shape == "s=1024, 5 k":
        kmers, s64, n = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32), 16, 1500
    elif shape == "default sketch, 5 k":
        kmers, s64, n = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32), 156, 700
    else:
        kmers, s64, n = K_CORONA, 156, 600
    sk, member = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=14, cluster_size=25, seed=77)
    sk[5] = sk[4]
    clu = (member % 2).astype(np.uint16)
    tbl = _table(kmers, n_clu=2)
    db = engine.SketchDB(sk, s64, 14, clusters=clu)
    name = lambda: engine._lib.lib().ppk_last_kernel_name().decode()
    ppk_option("ksplit", 0)
    base, fb = engine.dist(db, None, kmers, tbl)
    assert not name().endswith("fused>") and not name().endswith("parts>")
    want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, clu, threads=THREADS)
    assert int(fb.item()) == wf and np.abs(base.cpu().numpy() - want).max() <= TOL
    x_max, y_max = synth.boundary_for_quantile(want, 0.05)
    e_base, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
    ppk_option("ksplit", 1200)
    ppk_option("ks_grid_pad", 1)
    for slices in (0, 1, 2):
        ppk_option("ksplit_slices", slices)
        for rep in range(3):
            got, fg = engine.dist(db, None, kmers, tbl)
            assert name().endswith("fused>") or name().endswith("parts>"), name()
            assert torch_equal_bits(got, base) and int(fg.item()) == int(fb.item()), (slices, rep)
            e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
            assert np.array_equal(e.cpu().numpy(), e_base.cpu().numpy()), (slices, rep)
    db.close()


def test_long_sketch_rule_is_a_switch_and_changes_no_bit(ppk_option):
    """sketchsize64 32, 4 000 genomes (1 125 pair tiles: beyond every tile-count threshold): "ksplit_long" 1 (default)
    runs the k-split path, 0 the tile kernel; distances and the fused edge list agree bit for bit."""
    k5 = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
    sk, _ = synth.make_sketches(4000, k5, sketchsize64=32, bbits=14, cluster_size=40, seed=8)
    tbl = _table(k5)
    db = engine.SketchDB(sk, 32, 14)
    name = lambda: engine._lib.lib().ppk_last_kernel_name().decode()
    a, fa = engine.dist(db, None, k5, tbl)
    assert "k-split" in name()
    x_max, y_max = synth.boundary_for_quantile(a[:200000].cpu().numpy(), 0.05)
    ea, _ = engine.dist_edges(db, None, k5, tbl, slope=2, x_max=x_max, y_max=y_max)
    ppk_option("ksplit_long", 0)
    b, fb = engine.dist(db, None, k5, tbl)
    assert name().endswith("lds-dma>")
    eb, _ = engine.dist_edges(db, None, k5, tbl, slope=2, x_max=x_max, y_max=y_max)
    assert torch_equal_bits(a, b) and int(fa.item()) == int(fb.item())
    import torch
    assert torch.equal(ea, eb) and len(ea) > 1000
    db.close()


def test_wide_list_from_database_files_through_the_surface_popPUNK_calls(tmp_path):
    """A documented wide list (k = 13..31 step 2: ten lengths) at the default sketch size, from reference-layout `.h5` databases
    through `sketchlib.queryDatabase` (PopPUNK/sketchlib.py:475-632): self with the --plot-fit leg (ten Jaccards per
    example pair) and ref x query; the fused edge call from the same files."""
    import os
    from poppunk_amd import h5lite, sketchdb, sketchlib
    if not h5lite.available():
        pytest.skip("libhdf5 not found")
    kmers, s64 = K_STEP2, 156
    sk, _ = synth.make_sketches(230, kmers, sketchsize64=s64, bbits=14, cluster_size=23, seed=4)
    rn, qn = ["ref_%03d" % i for i in range(180)], ["qry_%02d" % i for i in range(50)]
    tbl = _table(kmers)
    rp, qp = str(tmp_path / "refdb"), str(tmp_path / "qrydb")
    sketchdb.save_h5(rp + "/refdb", rn, kmers, sk[:180], s64, 14, random_table=tbl, clusters=np.zeros(180, dtype=np.uint16))
    sketchdb.save_h5(qp + "/qrydb", qn, kmers, sk[180:], s64, 14)
    pp_sketchlib.clear_cache()
    assert list(sketchdb.getKmersFromReferenceDatabase(rp)) == kmers.tolist() and sketchdb.getSketchSize(rp)[0] == s64
    d = sketchlib.queryDatabase(rn, rn, rp, rp, kmers, self=True, number_plot_fits=1)
    want, wf = oracle.query(sk[:180], None, kmers, s64, 14, tbl, threads=THREADS)
    assert wf == 0 and d.shape == (16110, 2) and np.abs(d - want).max() <= TOL
    lines = open(rp + "/refdb_fit_example_1.tsv").read().strip().split("\n")
    assert len(lines) == 3 + len(kmers)
    d2 = sketchlib.queryDatabase(rn, qn, rp, qp, kmers, self=False)
    want2, _ = oracle.query(sk[:180], sk[180:], kmers, s64, 14, tbl, threads=THREADS)
    assert np.abs(d2 - want2).max() <= TOL
    x_max, y_max = synth.boundary_for_quantile(d, 0.1)
    e = pp_sketchlib.queryDatabaseEdges(rp + "/refdb", rp + "/refdb", rn, rn, kmers, 2, x_max, y_max, inclusive=True)
    assert np.array_equal(e, oracle.edge_threshold(d, 2, x_max, y_max))
    pp_sketchlib.clear_cache()


@pytest.mark.parametrize("s64,kmers,n", [(16, np.arange(3, 102), 420), (156, np.arange(3, 102, 2), 120)])
def test_the_longest_k_lists_the_reference_accepts(ppk_option, s64, kmers, n):
    """PopPUNK takes k = 3 .. 101 (PopPUNK/__main__.py: min-k >= 3, max-k <= 101, k-step >= 1): 99 k-mer lengths at most --
    nine windows of the count register at s = 1 024.  Distances by both routes, fused edges, neighbours."""
    kmers = kmers.astype(np.int32)
    sk, _ = synth.make_sketches(n, kmers, sketchsize64=s64, bbits=14, cluster_size=20, seed=1)
    tbl = (np.zeros((len(kmers), 1, 1)) + 1e-4).astype(np.float32)
    want, wf = oracle.query(sk, None, kmers, s64, 14, tbl, threads=THREADS)
    db = engine.SketchDB(sk, s64, 14)
    for ks in (1200, 0):
        ppk_option("ksplit", ks)
        d, f = engine.dist(db, None, kmers, tbl)
        assert engine._lib.lib().ppk_last_kernel_name().decode().endswith("wide>" if ks == 0 else "parts>")
        got = d.cpu().numpy()
        assert int(f.item()) == wf and np.abs(got - want).max() <= TOL
        x_max, y_max = synth.boundary_for_quantile(got, 0.1)
        e, _ = engine.dist_edges(db, None, kmers, tbl, slope=2, x_max=x_max, y_max=y_max)
        assert np.array_equal(e.cpu().numpy(), oracle.edge_threshold(got, 2, x_max, y_max))
    gi, gj, gd = (x.cpu().numpy() for x in engine.knn_from_sketches(db, kmers, tbl, 4, method="tiles"))
    wi, wj, wd = oracle.knn(oracle.long_to_square(got[:, 0]), 4)
    assert np.array_equal(gj, wj) and np.array_equal(gd, wd)
    db.close()
