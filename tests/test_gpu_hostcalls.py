"""The host-buffer layer of libppk_hip.so on a real MI355X: what stays on the device between
calls (fit tables, resident databases, parked results) must never answer for inputs it was not
computed from, and a device list runs its entries side by side (one worker thread each).
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from oracle import oracle
from poppunk_amd import _lib, engine, poppunk_refine, pp_sketchlib, synth

pytestmark = pytest.mark.gpu

KMERS = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)
TOL = 1e-6


def _stats():
    v = (C.c_double * 8)()
    _lib.check(_lib.lib().ppk_query_last_stats(v, 8), "ppk_query_last_stats")
    return dict(zip(("parts", "threads", "dl_max", "up_max", "wall_ms", "upload_ms_max", "part_ms_max", "reruns"),
                    list(v)))


def test_fit_tables_are_rebuilt_after_release_scratch():
    """ppk_release_scratch frees the block that holds the log-J / (E, F) tables; the next call with
    the same k list and table gets, as a rule, the same address back from hipMalloc -- the tables
    must be rebuilt, not taken for valid (round-2 advisor finding)."""
    sk, _ = synth.make_sketches(300, KMERS, cluster_size=30)
    tbl = synth.random_match_table(KMERS)
    want, wf = oracle.query(sk, None, KMERS, 16, 14, tbl, threads=4)
    lib = _lib.lib()
    for _ in range(3):
        got, gf = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
        assert gf == wf and np.abs(got - want).max() <= TOL
        lib.ppk_release_scratch()
        # something else takes (and scribbles over) the freed block before the tables come back
        junk = np.full((50000, 2), 0.5, dtype=np.float32)
        poppunk_refine.assignThreshold(junk, 2, 0.3, 0.3)
    db = engine.SketchDB(sk, 16, 14)
    for _ in range(2):
        d, _ = engine.dist(db, None, KMERS, tbl)
        assert np.abs(d.cpu().numpy() - want).max() <= TOL
        pp_sketchlib.clear_cache()
    db.close()


def test_in_place_rewrite_of_one_sample_at_70k_genomes_is_seen():
    """ppk_query keeps resident databases keyed by the host pointer, the dimensions and a hash of
    EVERY word.  Round 2 sampled 2^16 words: above ~65 000 genomes a one-sample change could go
    unnoticed and the call answered with the OLD sketches.  70 000 refs x 64 queries, default
    options: rewrite one ref in place (same pointer, same shape) -> the new counts.  (The call starts
    on the resident copy while the hash is still being computed and checks it before it returns: a
    stale copy costs a second run, never a wrong answer.)"""
    rng = np.random.Generator(np.random.PCG64(11))
    n_ref, n_qry = 70000, 64
    ref = rng.integers(0, 1 << 63, size=(n_ref, 5, 224), dtype=np.int64).astype(np.uint64)
    qry = rng.integers(0, 1 << 63, size=(n_qry, 5, 224), dtype=np.int64).astype(np.uint64)
    assert _lib.get_option("db_cache") == 1
    first, _ = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, counts=True)
    assert first.max() < 64                               # unrelated words: a handful of chance matches
    reruns0 = _stats()["reruns"]
    again, _ = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, counts=True)
    assert np.array_equal(first, again)
    assert _stats()["reruns"] == reruns0                  # unchanged arrays: the resident copies were right
    for n_seen, victim in enumerate((41234, 69999, 0), 1):
        ref[victim] = qry[7]                              # in place: every bin of every k now matches query 7
        got, _ = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, counts=True)
        assert np.array_equal(got[7 * n_ref + victim], np.full(5, 1024)), victim
        # the call started on the resident copy, found its hash stale before returning, and ran again
        assert _stats()["reruns"] == reruns0 + n_seen
    rows = np.concatenate([np.arange(7 * n_ref, 8 * n_ref), np.arange(0, n_ref)])
    want = oracle.match_counts(ref, qry[[7, 0]], 16, 14, threads=8)
    assert np.array_equal(got[rows], np.concatenate([want[:n_ref], want[n_ref:]]))
    # one WORD changed (the smallest possible edit)
    ref[12345, 3, 100] ^= np.uint64(1)
    got2, _ = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, counts=True)
    want2 = oracle.match_counts(ref[12345:12346], qry, 16, 14, threads=1)
    assert np.array_equal(got2[12345::n_ref], want2)
    pp_sketchlib.clear_cache()


def test_resident_database_cache_stays_bounded_and_correct():
    """Six different host arrays in turn (more than the four kept per device), each queried twice:
    every answer is its own."""
    tbl = synth.random_match_table(KMERS)
    arrays = [synth.make_sketches(150 + 10 * i, KMERS, cluster_size=25, seed=100 + i)[0] for i in range(6)]
    want = [oracle.query(a, None, KMERS, 16, 14, tbl, threads=4)[0] for a in arrays]
    for rnd in range(2):
        for a, w in zip(arrays, want):
            got, _ = pp_sketchlib.query_arrays(a, None, KMERS, 16, 14, tbl)
            assert np.abs(got - w).max() <= TOL
    pp_sketchlib.clear_cache()


def test_parked_result_is_fetched_explicitly_and_never_matched_to_a_later_call():
    """A host call whose buffer is too small parks the finished list for ppk_parked_fetch; the same
    entry point called again -- same pointer, same arguments, rewritten contents -- recomputes."""
    lib = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(5))
    samples = 300
    d = rng.random((samples * (samples - 1) // 2, 2)).astype(np.float32)
    fp = d.ctypes.data_as(C.POINTER(C.c_float))
    ll = C.POINTER(C.c_longlong)
    n = C.c_size_t(0)
    want = oracle.edge_threshold(d, 2, 0.5, 0.5)
    assert lib.ppk_edge_threshold(fp, d.shape[0], 0, 2, 0.5, 0.5, 1, 0, None, 0, C.byref(n)) == _lib.ERR_CAPACITY
    assert n.value == len(want) > 0
    # another thread has nothing parked
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.ppk_parked_fetch(None, None, None, 0, None)))
    t.start()
    t.join()
    assert seen == [_lib.ERR_STATE]
    ij = np.empty((n.value, 2), dtype=np.int64)
    m = C.c_size_t(0)
    assert lib.ppk_parked_fetch(ij.ctypes.data_as(ll), None, None, 3, C.byref(m)) == _lib.ERR_CAPACITY and m.value == n.value
    assert lib.ppk_parked_fetch(ij.ctypes.data_as(ll), None, None, n.value, C.byref(m)) == _lib.OK
    assert np.array_equal(ij, want)
    assert lib.ppk_parked_fetch(ij.ctypes.data_as(ll), None, None, n.value, None) == _lib.ERR_STATE   # fetched once
    # the size query that is never followed by a fetch (an exception between the two calls), then the same
    # array rewritten in place
    assert lib.ppk_edge_threshold(fp, d.shape[0], 0, 2, 0.5, 0.5, 1, 0, None, 0, C.byref(n)) == _lib.ERR_CAPACITY
    d[:] = rng.random(d.shape).astype(np.float32)
    want2 = oracle.edge_threshold(d, 2, 0.5, 0.5)
    assert not np.array_equal(want2, want)
    ij2 = np.empty((len(want2) + 5, 2), dtype=np.int64)
    assert lib.ppk_edge_threshold(fp, d.shape[0], 0, 2, 0.5, 0.5, 1, 0, ij2.ctypes.data_as(ll), len(ij2),
                                  C.byref(n)) == _lib.OK
    assert n.value == len(want2) and np.array_equal(ij2[:n.value], want2)
    assert lib.ppk_parked_fetch(ij.ctypes.data_as(ll), None, None, len(ij), None) == _lib.ERR_STATE   # dropped
    # the sweeps' three arrays
    offsets = np.linspace(-0.2, 0.3, 40) * np.sqrt(2)
    wi, wj, wo = oracle.threshold_iterate_1d(d, offsets, 2, 0.2, 0.2, 0.3, 0.3)
    od = offsets.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.ppk_threshold_iterate_1d(fp, d.shape[0], od, len(offsets), 2, 0.2, 0.2, 0.3, 0.3, 0, None, None,
                                        None, 0, C.byref(n)) == _lib.ERR_CAPACITY
    assert n.value == len(wi) > 0
    gi, gj, go = (np.empty(n.value, dtype=np.int64) for _ in range(3))
    assert lib.ppk_parked_fetch(gi.ctypes.data_as(ll), gj.ctypes.data_as(ll), go.ctypes.data_as(ll), n.value,
                                None) == _lib.OK
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(go, wo)
    # and the Python mirrors (guess too small -> fetch) on the rewritten matrix
    assert np.array_equal(poppunk_refine.edgeThreshold_array(d, 2, 0.9, 0.9), oracle.edge_threshold(d, 2, 0.9, 0.9))
    lib.ppk_release_scratch()


def test_device_list_entries_run_side_by_side(ppk_option):
    """ppk_query with a device list: one worker thread per entry uploads / computes / downloads its
    share.  On a 1-GPU box the same device is listed twice (separate streams and buffers): the result
    equals the one-device result bit for bit, two threads ran, and at some moment both had a
    download in flight (library counters)."""
    tbl = synth.random_match_table(KMERS)
    sk = synth.make_sketches(5000, KMERS, cluster_size=50, seed=9)[0]          # 12.5 M pairs, 100 MB of result
    one, f1 = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl, devices=(0,))
    st = _stats()
    assert st["parts"] == 1 and st["threads"] == 0 and st["dl_max"] == 1
    ppk_option("chunk_rows", 800000)                                           # ~8 sub-bands of 6.4 MB per entry
    overlapped = 0
    for _ in range(12):
        if overlapped == 2:
            break
        two, f2 = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl, devices=(0, 0))
        assert f2 == f1 and np.array_equal(one, two)
        st = _stats()
        assert st["parts"] == 2 and st["threads"] == 2
        overlapped = max(overlapped, st["dl_max"])
    assert overlapped == 2
    # cold start with the cache off: the leader entry uploads, the other waits for it
    ppk_option("db_cache", 0)
    three, f3 = pp_sketchlib.query_arrays(sk[:3000], sk[3000:], KMERS, 16, 14, tbl, devices=(0, 0, 0))
    base, fb = pp_sketchlib.query_arrays(sk[:3000], sk[3000:], KMERS, 16, 14, tbl, devices=(0,))
    assert f3 == fb and np.array_equal(three, base)
    want, _ = oracle.query(sk[:3000], sk[3000:3200], KMERS, 16, 14, tbl, threads=8)
    assert np.abs(base[:3000 * 200] - want).max() <= TOL
    pp_sketchlib.clear_cache()


def test_queryDatabase_holds_handles_and_splits_over_PPK_DEVICES(tmp_path, monkeypatch):
    """The Python mirror keeps ppk_db handles with its loaded databases and calls ppk_query_dbs:
    self, ref x query and sub-sample re-queries, one device and a list; a rewritten FILE is read
    again (the key holds its modification time)."""
    from poppunk_amd import sketchdb
    sk, _ = synth.make_sketches(700, KMERS, cluster_size=35, seed=21)
    tbl = synth.random_match_table(KMERS)
    names = ["s%04d" % i for i in range(700)]
    db = str(tmp_path / "db")
    sketchdb.save_npz(db, names, KMERS, sk, 16, 14, random_table=tbl)
    pp_sketchlib.clear_cache()
    klist = KMERS.tolist()
    want, _ = oracle.query(sk, None, KMERS, 16, 14, tbl, threads=4)
    got = pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, 0)
    assert np.abs(got - want).max() <= TOL
    assert len(pp_sketchlib._DB_CACHE) == 1
    entry = next(iter(pp_sketchlib._DB_CACHE.values()))
    assert len(entry._handles) == 1
    # --plot-fit style re-queries of single samples: sliced from the loaded database, never cached
    for a, b in ((3, 4), (10, 699)):
        j = pp_sketchlib.queryDatabase(db, db, [names[a]], [names[b]], klist, True, True, 1, True, 0)
        wj = oracle.query(sk[a:a + 1], sk[b:b + 1], KMERS, 16, 14, tbl, jaccard=True, threads=1)[0]
        assert np.array_equal(j, wj)
    assert len(pp_sketchlib._DB_CACHE) == 1 and len(entry._handles) == 1
    # ref x query out of one file, over a device list
    monkeypatch.setenv("PPK_DEVICES", "0,0")
    rq = pp_sketchlib.queryDatabase(db, db, names[:500], names[500:], klist, True, False, 1, True, 0)
    assert np.abs(rq - oracle.query(sk[:500], sk[500:], KMERS, 16, 14, tbl, threads=4)[0]).max() <= TOL
    assert _stats()["threads"] == 2
    again = pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, 0)
    assert np.array_equal(again, got)
    monkeypatch.delenv("PPK_DEVICES")
    # the file is rewritten with other sketches under the same names
    sk2, _ = synth.make_sketches(700, KMERS, cluster_size=35, seed=22)
    sketchdb.save_npz(db, names, KMERS, sk2, 16, 14, random_table=tbl)
    os.utime(db + ".npz", ns=(1, 1))                      # whatever the clock granularity
    got2 = pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, 0)
    assert np.abs(got2 - oracle.query(sk2, None, KMERS, 16, 14, tbl, threads=4)[0]).max() <= TOL
    pp_sketchlib.clear_cache()
    assert not pp_sketchlib._DB_CACHE


def test_pin_kit_runs_its_own_half_without_upstream(tmp_path):
    """tools/pin_upstream.py settles DESIGN.md section 5's [EXT] rows wherever upstream pp-sketchlib is
    installed.  Here it is not: the script must say so, still write its databases and run this
    package's half of every comparison, and exit 2 ("nothing pinned")."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        import pp_sketchlib as upstream  # noqa: F401
        pytest.skip("upstream pp-sketchlib is importable here: run tools/pin_upstream.py itself")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pin_upstream.py"), "--ours-only", "--out",
                        str(tmp_path), "--keep"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, r.stdout[-3000:] + r.stderr[-3000:]
    assert "NOTHING PINNED" in r.stdout and "our half ran" in r.stdout
    for tag in ("s1024", "s19200", "s1024q", "gap"):
        assert os.path.exists(str(tmp_path / (tag + ".h5")))
    assert "our two readings differ on 6 of 66 rows" in r.stdout      # the gap pairs tell truncate from skip


def test_pin_kit_compares_the_refine_extension_function_by_function(tmp_path):
    """The kit's first step compares upstream `poppunk_refine` with the mirrors.  Upstream is absent here, so a
    stand-in of that name built on the oracle is put on the path: the step must find it, run all of its
    comparisons (ties everywhere in its inputs) and report every function identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "poppunk_refine.py").write_text('''
import numpy as np
from oracle import oracle as o
def _t(e): return [tuple(r) for r in np.asarray(e).reshape(-1, 2).tolist()]
def _l(t): return tuple(np.asarray(x).tolist() for x in t)
def assignThreshold(d, slope, x, y, num_threads=1): return o.assign_threshold(d, slope, x, y)
def edgeThreshold(d, slope, x, y): return _t(o.edge_threshold(d, slope, x, y))
def generateTuples(a, label, self=True, num_ref=0, int_offset=0): return _t(o.generate_tuples(a, label, self, num_ref, int_offset))
def generateAllTuples(num_ref, num_queries=0, self=True, int_offset=0): return _t(o.generate_all_tuples(num_ref, num_queries, self, int_offset))
def thresholdIterate1D(d, offsets, slope, x0, y0, x1, y1, num_threads=1): return _l(o.threshold_iterate_1d(d, offsets, slope, x0, y0, x1, y1))
def thresholdIterate2D(d, x_max, y_max): return _l(o.threshold_iterate_2d(d, x_max, y_max))
def get_kNN_distances(sq, k, dist_col=0, num_threads=1): return _l(o.knn(sq, k))
def lowerRank(m, n, k, recip, unique, eps, num_threads=1): return _l(o.lower_rank(m[0], m[1], m[2], n, k, recip, unique, eps))
def extend(m, qq, qr, k, num_threads=1): return _l(o.extend(m[0], m[1], m[2], qq, qr, k))
''')
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pin_upstream.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert "every function identical" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("identical") >= 25 and "DIFFERS" not in r.stdout
    assert r.returncode == 2          # pp_sketchlib itself is still absent: kernel 1 stays unpinned


@pytest.mark.parametrize("n,threads", [(460, 8), (3750, 8), (3750, 3), (7200, 16), (3750, 0)])
def test_staged_uploads_of_pageable_arrays(ppk_option, n, threads):
    """Host arrays reach the device through a pinned ring filled by helper threads (32 MB pieces, two slots):
    below the staging threshold (4 MB), one piece and a ragged second one (33.6 MB), three pieces (64.5 MB),
    any helper count, and helpers switched off (the runtime's own path) -- the resident database is the same
    bytes: match counts against the oracle, and the kernel-2 host calls on a 36 MB matrix."""
    ppk_option("prefault_threads", threads)
    pp_sketchlib.clear_cache()
    sk = synth.make_sketches(n, KMERS, cluster_size=50, seed=70 + n)[0]
    assert sk.nbytes == n * 8960
    q = sk[:7]
    got, _ = pp_sketchlib.query_arrays(sk, q, KMERS, 16, 14, counts=True)
    assert np.array_equal(got, oracle.match_counts(sk, q, 16, 14, threads=8))
    rng = np.random.Generator(np.random.PCG64(n))
    m = 3000
    d = rng.random((m * (m - 1) // 2, 2)).astype(np.float32)          # 36 MB: two ring pieces
    assert np.array_equal(poppunk_refine.assignThreshold(d, 2, 0.4, 0.5), oracle.assign_threshold(d, 2, 0.4, 0.5))
    assert np.array_equal(poppunk_refine.edgeThreshold_array(d, 2, 0.1, 0.1), oracle.edge_threshold(d, 2, 0.1, 0.1))
    pp_sketchlib.clear_cache()


# ---- sketches -> edge list as one host call (ppk_query_edges / ppk_query_edges_dbs) ------------------

@pytest.mark.parametrize("devices", [(0,), (0, 0, 0)])
@pytest.mark.parametrize("inclusive", [False, True])
def test_query_edges_host_call_equals_two_step(devices, inclusive):
    """The fused host call gives the list of queryDatabase -> X / scale -> edgeThreshold (inclusive) or
    assignThreshold -> generateTuples (strict) on the SAME distances, whatever the device list: bands of
    rows, concatenated in row order."""
    sk, _ = synth.make_sketches(900, KMERS, cluster_size=30, seed=5)
    tbl = synth.random_match_table(KMERS)
    dist, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    odist, _ = oracle.query(sk, None, KMERS, 16, 14, tbl, threads=4)
    assert np.abs(dist - odist).max() <= TOL
    x_max, y_max = synth.boundary_for_quantile(dist, 0.08)
    scale = (np.float32(dist[:, 0].max()), np.float32(dist[:, 1].max()))
    scaled = np.ascontiguousarray(dist / np.asarray(scale, dtype=np.float32))      # PopPUNK/models.py:1085
    xs, ys = x_max / float(scale[0]), y_max / float(scale[1])
    for slope in (0, 1, 2):
        want = oracle.edge_threshold(scaled, slope, xs, ys, inclusive=inclusive)
        assert len(want) > 100
        got, nf = pp_sketchlib.query_edges_arrays(sk, None, KMERS, 16, 14, slope, xs, ys, scale=scale,
                                                  inclusive=inclusive, random_table=tbl, devices=devices)
        assert nf == 0 and got.dtype == np.int64 and np.array_equal(got, want)
        # too little room: the finished list is parked on the host and fetched, not recomputed
        got2, _ = pp_sketchlib.query_edges_arrays(sk, None, KMERS, 16, 14, slope, xs, ys, scale=scale,
                                                  inclusive=inclusive, random_table=tbl, devices=devices, cap=7)
        assert np.array_equal(got2, want)
    # the strict list is what the two library calls PopPUNK makes give (models.py:1088, network.py:1180)
    if not inclusive:
        a = poppunk_refine.assignThreshold(scaled, 2, xs, ys, 1)
        t = poppunk_refine.generateTuples(a.astype(np.int32).tolist(), -1, self=True, num_ref=0, int_offset=0)
        got, _ = pp_sketchlib.query_edges_arrays(sk, None, KMERS, 16, 14, 2, xs, ys, scale=scale,
                                                 inclusive=False, random_table=tbl, devices=devices)
        assert [tuple(r) for r in got.tolist()] == [tuple(x) for x in t]


def test_query_edges_works_through_a_band_in_pieces(ppk_option):
    """The edge bitmask of a device's band is bounded: with a small piece size the same lists come out of
    many passes (self and ref x query, one entry and three, buffer regrown between pieces)."""
    sk, _ = synth.make_sketches(1100, KMERS, cluster_size=25, seed=15)
    tbl = synth.random_match_table(KMERS)
    dist, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    x_max, y_max = synth.boundary_for_quantile(dist, 0.1)
    want = oracle.edge_threshold(dist, 2, x_max, y_max, inclusive=True)
    rq, _ = pp_sketchlib.query_arrays(sk[:700], sk[700:], KMERS, 16, 14, tbl)
    want_rq = oracle.edge_threshold(rq, 2, x_max, y_max, n_ref=700, inclusive=True)
    assert len(want) > 1000 and len(want_rq) > 500
    ppk_option("chunk_rows", 64)          # pieces of 64 * 32 mask words: 64 - 128 query rows each
    for devices in ((0,), (0, 0, 0)):
        got, _ = pp_sketchlib.query_edges_arrays(sk, None, KMERS, 16, 14, 2, x_max, y_max, random_table=tbl,
                                                 devices=devices)
        assert np.array_equal(got, want)
        got, _ = pp_sketchlib.query_edges_arrays(sk[:700], sk[700:], KMERS, 16, 14, 2, x_max, y_max,
                                                 random_table=tbl, devices=devices)
        assert np.array_equal(got, want_rq)


def test_query_edges_ref_query_clusters_and_errors():
    sk, clu = synth.make_sketches(640, KMERS, cluster_size=20, seed=8)
    tbl = synth.random_match_table(KMERS)
    ref, qry = sk[:500], sk[500:]
    dist, _ = pp_sketchlib.query_arrays(ref, qry, KMERS, 16, 14, tbl)
    x_max, y_max = synth.boundary_for_quantile(dist, 0.1)
    want = oracle.edge_threshold(dist, 2, x_max, y_max, n_ref=500, inclusive=True)
    assert len(want) > 100
    for devices in ((0,), (0, 0)):
        got, _ = pp_sketchlib.query_edges_arrays(ref, qry, KMERS, 16, 14, 2, x_max, y_max, random_table=tbl,
                                                 devices=devices)
        assert np.array_equal(got, want)
    # an empty list, and a boundary nothing lies outside
    none, _ = pp_sketchlib.query_edges_arrays(ref, qry, KMERS, 16, 14, 2, 1e-9, 1e-9, random_table=tbl,
                                              inclusive=False)
    assert none.shape == (0, 2)
    every, _ = pp_sketchlib.query_edges_arrays(ref, qry, KMERS, 16, 14, 2, 50.0, 50.0, random_table=tbl,
                                               devices=(0, 0))
    assert len(every) == 500 * 140 and np.array_equal(every[:3], [[0, 500], [1, 500], [2, 500]])
    # a thread that parked nothing has nothing to fetch, and a fetched list is gone
    lib = _lib.lib()
    buf = np.empty((4, 2), dtype=np.int64)
    n = C.c_size_t(0)
    assert lib.ppk_parked_fetch(buf.ctypes.data_as(C.POINTER(C.c_longlong)), None, None, 4, C.byref(n)) == _lib.ERR_STATE
    with pytest.raises(RuntimeError, match="slope"):
        pp_sketchlib.query_edges_arrays(ref, qry, KMERS, 16, 14, 3, x_max, y_max, random_table=tbl)
    with pytest.raises(RuntimeError, match="at most 4"):
        pp_sketchlib.query_edges_arrays(ref, qry, KMERS, 16, 14, 2, x_max, y_max, random_table=tbl,
                                        devices=(0,) * 5)


def test_queryDatabaseEdges_mirror(tmp_path, monkeypatch):
    """The database-file form: the loaded database's handles, PPK_DEVICES, and a model's scale."""
    from poppunk_amd import sketchdb
    sk, _ = synth.make_sketches(800, KMERS, cluster_size=40, seed=31)
    tbl = synth.random_match_table(KMERS)
    names = ["g%04d" % i for i in range(800)]
    db = str(tmp_path / "db")
    sketchdb.save_npz(db, names, KMERS, sk, 16, 14, random_table=tbl)
    pp_sketchlib.clear_cache()
    klist = KMERS.tolist()
    dist = pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, 0)
    x_max, y_max = synth.boundary_for_quantile(dist, 0.05)
    scale = (np.float32(dist[:, 0].max()), np.float32(dist[:, 1].max()))
    scaled = np.ascontiguousarray(dist / np.asarray(scale, dtype=np.float32))
    xs, ys = x_max / float(scale[0]), y_max / float(scale[1])
    want = oracle.edge_threshold(scaled, 2, xs, ys, inclusive=False)
    assert len(want) > 100
    got = pp_sketchlib.queryDatabaseEdges(db, db, names, names, klist, 2, xs, ys, scale=scale)
    assert np.array_equal(got, want)
    monkeypatch.setenv("PPK_DEVICES", "0,0")
    got = pp_sketchlib.queryDatabaseEdges(db, db, names, names, klist, 2, xs, ys, scale=scale)
    assert np.array_equal(got, want)
    # ref x query out of the same file (transient sub-sample entries)
    rq = pp_sketchlib.queryDatabase(db, db, names[:600], names[600:], klist, True, False, 1, True, 0)
    wq = oracle.edge_threshold(rq, 2, x_max, y_max, n_ref=600, inclusive=True)
    gq = pp_sketchlib.queryDatabaseEdges(db, db, names[:600], names[600:], klist, 2, x_max, y_max, inclusive=True)
    assert len(wq) > 50 and np.array_equal(gq, wq)
    pp_sketchlib.clear_cache()


# ---- sketches -> k nearest neighbours as one host call (ppk_query_knn / ppk_query_knn_dbs) -----------------

@pytest.mark.parametrize("devices", [(0,), (0, 0, 0)])
def test_query_knn_host_call(devices):
    """Every device takes a band of the triangle, the bands' best-k lists are merged on the host: the result
    is get_kNN_distances(longToSquare(distances)) whatever the device list -- ties (unrelated clusters: most
    distances equal) included."""
    for n, cluster_size, related in ((900, 30, True), (700, 700, False), (5, 5, True)):
        sk, _ = synth.make_sketches(n, KMERS, cluster_size=cluster_size, seed=n, related=related)
        tbl = synth.random_match_table(KMERS)
        dist, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
        for knn, col in ((1, 0), (7, 1), (32, 0)):
            wi, wj, wd = oracle.knn(oracle.long_to_square(dist[:, col]), knn)
            gi, gj, gd = pp_sketchlib.query_knn_arrays(sk, KMERS, 16, 14, knn, col, tbl, devices=devices)
            assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gd, wd), (n, knn, col)
    with pytest.raises(RuntimeError, match=r"knn must be in \[1, 32\]"):
        pp_sketchlib.query_knn_arrays(sk, KMERS, 16, 14, 33, 0, tbl, devices=devices)


def test_queryDatabaseKNN_mirror(tmp_path, monkeypatch):
    from poppunk_amd import sketchdb
    sk, _ = synth.make_sketches(600, KMERS, cluster_size=24, seed=41)
    tbl = synth.random_match_table(KMERS)
    names = ["k%04d" % i for i in range(600)]
    db = str(tmp_path / "db")
    sketchdb.save_npz(db, names, KMERS, sk, 16, 14, random_table=tbl)
    pp_sketchlib.clear_cache()
    klist = KMERS.tolist()
    dist = pp_sketchlib.queryDatabase(db, db, names, names, klist, True, False, 1, True, 0)
    want = poppunk_refine.get_kNN_distances(pp_sketchlib.longToSquare(np.ascontiguousarray(dist[:, 0])), 4)
    for env in (None, "0,0"):
        if env:
            monkeypatch.setenv("PPK_DEVICES", env)
        i, j, d = pp_sketchlib.queryDatabaseKNN(db, names, klist, 4)
        assert i.tolist() == list(want[0]) and j.tolist() == list(want[1])
        assert np.array_equal(d, np.asarray(want[2], dtype=np.float32))
    # a subset of the database (names in another order): neighbours within the subset
    sub = names[100:400][::-1]
    ds = pp_sketchlib.queryDatabase(db, db, sub, sub, klist, True, False, 1, True, 0)
    ws = oracle.knn(oracle.long_to_square(ds[:, 1]), 3)
    i, j, d = pp_sketchlib.queryDatabaseKNN(db, sub, klist, 3, dist_col=1)
    assert np.array_equal(j, ws[1]) and np.array_equal(d, ws[2])
    pp_sketchlib.clear_cache()


def test_repeated_host_calls_do_not_leak_device_memory():
    """Every host entry point of the round, 25 times over: after ppk_release_scratch the device has as much free
    memory as after the first pass (what stays is scratch, caches and resident databases -- all released)."""
    import torch
    from poppunk_amd import qc
    sk, _ = synth.make_sketches(600, KMERS, cluster_size=30, seed=2)
    tbl = synth.random_match_table(KMERS)
    dist, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    x_max, y_max = synth.boundary_for_quantile(dist, 0.1)
    sq = pp_sketchlib.longToSquare(np.ascontiguousarray(dist[:, 0]))
    qq, _ = pp_sketchlib.query_arrays(sk[500:], None, KMERS, 16, 14, tbl)
    qr, _ = pp_sketchlib.query_arrays(sk[:500], sk[500:], KMERS, 16, 14, tbl)

    def one_pass():
        pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl, devices=(0, 0))
        pp_sketchlib.query_edges_arrays(sk, None, KMERS, 16, 14, 2, x_max, y_max, random_table=tbl, devices=(0, 0), cap=5)
        pp_sketchlib.query_knn_arrays(sk, KMERS, 16, 14, 4, 0, tbl, devices=(0, 0))
        poppunk_refine.assignThreshold(dist, 2, x_max, y_max)
        poppunk_refine.edgeThreshold_array(dist, 2, x_max, y_max)
        poppunk_refine.thresholdIterate1D_arrays(dist, np.linspace(0, 0.2, 7), 2, 0.0, 0.0, x_max, y_max)
        poppunk_refine.generateAllTuples_array(300)
        qc.qc_edge_lists(dist, 0, 0.02, 0.3)
        nn = poppunk_refine.get_kNN_distances(sq, 6)
        poppunk_refine.lowerRank_arrays(nn, 600, 2, True, True, 1e-4)
        sub = poppunk_refine.get_kNN_distances(np.ascontiguousarray(sq[:500, :500]), 6)
        poppunk_refine.extend_arrays(sub, oracle.long_to_square(qq[:, 0]), np.ascontiguousarray(qr[:, 0].reshape(100, 500).T), 6)

    def free_bytes():
        pp_sketchlib.clear_cache()               # resident databases, query buffers, scratch
        _lib.lib().ppk_release_scratch()
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(0)[0]

    one_pass()
    base = free_bytes()
    for _ in range(25):
        one_pass()
    after = free_bytes()
    assert base - after < (64 << 20), "device memory shrank by %d MB over 25 passes" % ((base - after) >> 20)


def test_fused_host_edge_call_is_steady_at_100k_genomes():
    """BENCH_r03 recorded [271, 801, 271] ms for three calls of ppk_query_edges_dbs at 100 000 genomes.  Until round 4
    every call allocated its device edge list (rows / 8 entries: 10 GB here) and freed it again; the list is now kept
    per device entry (grow-only, tools/stall_hunt.py).  20 calls: none above 1.3 x the median, the buffer is
    allocated once, and ppk_release_scratch gives the memory back."""
    import time
    import torch
    n = 100000
    tbl = synth.random_match_table(KMERS)
    ref = engine.SketchDB(synth.make_sketches_device(n, KMERS, device="cuda:0"), 16, 14, device=0)
    sub = engine.SketchDB(synth.make_sketches_device(2000, KMERS, device="cuda:0"), 16, 14, device=0)
    d_sub, _ = engine.dist(sub, None, KMERS, tbl)
    x_max, y_max = synth.boundary_for_quantile(synth.tensor_to_numpy(d_sub), 0.02)
    sub.close()
    del d_sub
    _lib.lib().ppk_release_scratch()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    first, _ = engine.edges_host([ref], None, KMERS, tbl, slope=2, x_max=x_max, y_max=y_max, cap=16 << 20)
    held = free0 - torch.cuda.mem_get_info(0)[0]
    def batch():
        ms = []
        for _ in range(20):
            t0 = time.perf_counter()
            edges, _ = engine.edges_host([ref], None, KMERS, tbl, slope=2, x_max=x_max, y_max=y_max, cap=16 << 20)
            ms.append((time.perf_counter() - t0) * 1e3)
            assert len(edges) == len(first)
        assert np.array_equal(edges, first)
        return ms

    ms = batch()
    assert abs((free0 - torch.cuda.mem_get_info(0)[0]) - held) < (64 << 20)       # nothing allocated after call 1
    if max(ms) > 1.3 * float(np.median(ms)):
        ms = batch()          # one hiccup of the box is not the library's; a stall that comes back is
    med = float(np.median(ms))
    assert max(ms) <= 1.3 * med, "a stalled call: %s" % " ".join("%.0f" % x for x in ms)
    ref.close()
    _lib.lib().ppk_release_scratch()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info(0)[0] < (64 << 20) + n * 8960 * 0        # scratch and list released


def test_host_entry_points_from_four_threads_at_once():
    """The library is called from any thread (PopPUNK's refine optimiser runs thresholdIterate2D from a pool;
    a web service answers queries concurrently): four threads, each looping over a different group of entry
    points, get what a single thread gets."""
    from poppunk_amd import qc
    sk, _ = synth.make_sketches(500, KMERS, cluster_size=25, seed=12)
    tbl = synth.random_match_table(KMERS)
    dist, _ = pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)
    x_max, y_max = synth.boundary_for_quantile(dist, 0.1)
    sq = pp_sketchlib.longToSquare(np.ascontiguousarray(dist[:, 1]))

    jobs = [
        lambda: pp_sketchlib.query_arrays(sk, None, KMERS, 16, 14, tbl)[0],
        lambda: pp_sketchlib.query_edges_arrays(sk, None, KMERS, 16, 14, 2, x_max, y_max, random_table=tbl)[0],
        lambda: np.stack(pp_sketchlib.query_knn_arrays(sk, KMERS, 16, 14, 5, 1, tbl)[1:2]),
        lambda: poppunk_refine.edgeThreshold_array(dist, 2, x_max, y_max),
        lambda: np.stack(poppunk_refine.lowerRank_arrays(poppunk_refine.get_kNN_distances(sq, 7), 500, 3, True, False, 0.0)[:2]),
        lambda: qc.qc_edge_lists(dist, 0, 0.02, 0.3)[0],
        lambda: poppunk_refine.assignThreshold(dist, 1, x_max, y_max),
        lambda: np.stack(poppunk_refine.thresholdIterate2D_arrays(dist, np.linspace(0.01, x_max, 6), y_max)),
    ]
    want = [j() for j in jobs]
    errors = []

    def worker(t):
        try:
            for rep in range(6):
                for idx in range(t, len(jobs), 4):          # thread t: jobs t and t + 4
                    if not np.array_equal(jobs[idx](), want[idx]):
                        errors.append("thread %d job %d rep %d differs" % (t, idx, rep))
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:4]
