"""CPU tests of the host-side mirror of the reference interface: argument handling and error
behaviour of sketchlib.queryDatabase (PopPUNK/sketchlib.py:475-632), the noconvert rule of
poppunk_refine (src/python_bindings.cpp:82,:89), the sketch database files and the synthetic
generator.  No GPU compute is triggered here."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from poppunk_amd import poppunk_refine, pp_sketchlib, sketchdb, sketchlib, synth


def make_db(tmp_path, name, n, kmers=(13, 17, 21), with_random=True):
    sk, member = synth.make_sketches(n, kmers, cluster_size=4, seed=n)
    names = ["%s_%d" % (name, i) for i in range(n)]
    prefix = str(tmp_path / name)
    db_name = prefix + "/" + name
    sketchdb.save_npz(db_name, names, kmers, sk, 16, 14,
                      random_table=synth.random_match_table(kmers) if with_random else None,
                      clusters=np.zeros(n, dtype=np.uint16) if with_random else None)
    return prefix, names, sk


def test_self_query_requires_same_db(tmp_path):
    p1, n1, _ = make_db(tmp_path, "a", 6)
    p2, _, _ = make_db(tmp_path, "b", 6)
    with pytest.raises(RuntimeError, match="Must use same db for self query"):
        sketchlib.queryDatabase(n1, n1, p1, p2, [13, 17, 21], self=True)


def test_overlapping_names_exit(tmp_path, capsys):
    p1, n1, _ = make_db(tmp_path, "a", 6)
    with pytest.raises(SystemExit) as e:
        sketchlib.queryDatabase(n1, n1[:2], p1, p1, [13, 17, 21], self=False)
    assert e.value.code == 1
    assert "Unique names are required" in capsys.readouterr().err


def test_sketchdb_roundtrip_subset_and_order(tmp_path):
    prefix, names, sk = make_db(tmp_path, "db", 9)
    db_name = prefix + "/db"
    pick = [names[5], names[0], names[7]]
    got = sketchdb.load(db_name, pick, [21, 13])
    assert got.sketches.shape == (3, 2, 224)
    assert np.array_equal(got.sketches[0, 0], sk[5, 2]) and np.array_equal(got.sketches[1, 1], sk[0, 0])
    assert got.random_table.shape == (2, 1, 1) and got.clusters.shape == (3,)
    assert got.sketchsize64 == 16 and got.bbits == 14
    with pytest.raises(RuntimeError, match="not found"):
        sketchdb.load(db_name, ["nope"], [13])
    with pytest.raises(RuntimeError, match="k-mer length 15"):
        sketchdb.load(db_name, pick, [15])
    with pytest.raises(RuntimeError, match="not found"):
        sketchdb.load(str(tmp_path / "missing" / "missing"), pick, [13])


def test_query_arrays_argument_checks():
    sk, _ = synth.make_sketches(4, [13, 17], cluster_size=2)
    with pytest.raises(RuntimeError, match="klist"):
        pp_sketchlib.query_arrays(sk, None, [13, 17, 21], 16, 14)
    with pytest.raises(RuntimeError, match="sketchsize64"):
        pp_sketchlib.query_arrays(sk, None, [13, 17], 15, 14)
    with pytest.raises(RuntimeError, match="cluster ids"):
        pp_sketchlib.query_arrays(sk, None, [13, 17], 16, 14,
                                  random_table=np.zeros((2, 2, 2), dtype=np.float32))
    with pytest.raises(RuntimeError, match="out of range"):
        pp_sketchlib.query_arrays(sk, None, [13, 17], 16, 14,
                                  random_table=np.zeros((2, 2, 2), dtype=np.float32),
                                  ref_clusters=np.array([0, 1, 2, 0]))
    # a single sample has no pairs: empty result without touching the device
    out, failed = pp_sketchlib.query_arrays(sk[:1], None, [13, 17], 16, 14)
    assert out.shape == (0, 2) and failed == 0


def test_refine_noconvert_and_empty_inputs():
    with pytest.raises(TypeError):
        poppunk_refine.assignThreshold(np.zeros((4, 2)), 2, 0.5, 0.5)             # float64
    with pytest.raises(TypeError):
        poppunk_refine.assignThreshold(np.zeros((4, 3), dtype=np.float32), 2, 0.5, 0.5)
    with pytest.raises(TypeError):
        poppunk_refine.edgeThreshold([[0.0, 0.0]], 2, 0.5, 0.5)                   # list
    with pytest.raises(TypeError):
        poppunk_refine.generateTuples(np.zeros((2, 2)), -1)
    empty = np.zeros((0, 2), dtype=np.float32)
    assert poppunk_refine.assignThreshold(empty, 2, 0.5, 0.5).shape == (0,)
    assert poppunk_refine.edgeThreshold(empty, 2, 0.5, 0.5) == []
    assert poppunk_refine.generateTuples([], -1) == []


def test_iterDistRows_contract():
    names = ["a", "b", "c", "d"]
    assert list(sketchlib.iterDistRows(names, names, True)) == \
        [("b", "a"), ("c", "a"), ("d", "a"), ("c", "b"), ("d", "b"), ("d", "c")]
    assert list(sketchlib.iterDistRows(["r0", "r1"], ["q0", "q1", "q2"], False)) == \
        [("r0", "q0"), ("r1", "q0"), ("r0", "q1"), ("r1", "q1"), ("r0", "q2"), ("r1", "q2")]
    with pytest.raises(RuntimeError):
        list(sketchlib.iterDistRows(names, names[:2], True))


def test_synthetic_generator_properties():
    kmers = synth.DEFAULT_KMERS
    a, ma = synth.make_sketches(120, kmers, cluster_size=30)
    b, _ = synth.make_sketches(120, kmers, cluster_size=30)
    assert np.array_equal(a, b) and a.dtype == np.uint64 and a.shape == (120, 5, 224)
    tbl = synth.random_match_table(kmers)
    assert tbl.shape == (5, 1, 1) and np.all(np.diff(tbl.ravel()) < 0) and 0.02 < tbl[0, 0, 0] < 0.04
    assert np.isfinite(tbl).all()
    bins = np.arange(128, dtype=np.uint16).reshape(1, 128) * 101 % (1 << 14)
    w = synth.bitslice(bins, 14)
    assert w.shape == (1, 28)
    # word [blk*14 + b] bit i == bit b of bin 64*blk + i
    for blk in range(2):
        for bit in range(14):
            for i in (0, 1, 37, 63):
                assert ((int(w[0, blk * 14 + bit]) >> i) & 1) == ((int(bins[0, 64 * blk + i]) >> bit) & 1)


def test_dists_pickle_roundtrip_and_layout(tmp_path):
    """`.dists.pkl/.npy` as PopPUNK/utils.py:135-196 writes and reads them."""
    import pickle
    from poppunk_amd import distfile
    names = ["a", "b", "c"]
    X = np.arange(6, dtype=np.float32).reshape(3, 2)
    prefix = str(tmp_path / "db.dists")
    distfile.storePickle(names, names, True, X, prefix)
    with open(prefix + ".pkl", "rb") as f:
        assert pickle.load(f) == [names, names, True]            # exactly the reference's payload
    assert np.array_equal(np.load(prefix + ".npy"), X)
    r, q, s, Y = distfile.readPickle(prefix, enforce_self=True)
    assert (r, q, s) == (names, names, True) and np.array_equal(X, Y) and Y.dtype == np.float32
    assert distfile.readPickle(prefix, distances=False)[3] is None
    distfile.storePickle(names, ["q"], False, None, prefix)
    with pytest.raises(SystemExit):
        distfile.readPickle(prefix, enforce_self=True)


def test_clusters_from_edges():
    from poppunk_amd import distfile
    n_comp, labels = distfile.clusters_from_edges(6, [(0, 1), (1, 2), (4, 5)])
    assert n_comp == 3
    assert labels[0] == labels[1] == labels[2] and labels[4] == labels[5] and labels[3] not in (labels[0], labels[4])
    assert distfile.clusters_from_edges(3, np.zeros((0, 2), dtype=np.int64))[0] == 3


H5_PYTHON = "/opt/conda/bin/python3.9"      # this image's second interpreter is the one with h5py


def _h5_python_ok():
    import subprocess
    if not os.path.exists(H5_PYTHON):
        return False
    return subprocess.run([H5_PYTHON, "-c", "import h5py, numpy"], capture_output=True).returncode == 0


@pytest.mark.skipif(not _h5_python_ok(), reason="no interpreter with h5py in this image")
def test_reference_h5_layout_reader_writer_and_random_group(tmp_path):
    """The reference keeps sketches in <prefix>/<basename>.h5: group /sketches, one group per sample
    with attrs sketchsize64 / bbits / kmers / length / ..., one uint64 dataset per k named str(k)
    (PopPUNK/web.py:14-61), and pp-sketchlib's /random group beside it.  h5py (the image's conda
    interpreter) WRITES such a file -- with a /random group in the layout recalled from pp-sketchlib
    [EXT] -- and this interpreter reads it natively (h5lite over libhdf5): subset and order of names
    and of k, the mapped random table, the verbatim conversion to .npz; then this interpreter's
    writer produces a .h5 that h5py reads back value for value."""
    import subprocess
    from poppunk_amd import h5lite, sketchdb, synth
    if not h5lite.available():
        pytest.skip("libhdf5 not found")
    kmers = np.asarray([13, 17, 21, 25], dtype=np.int32)
    sk, _ = synth.make_sketches(9, kmers, sketchsize64=3, bbits=14, cluster_size=3, seed=12)
    names = ["s%02d" % i for i in range(9)]
    tbl = (np.random.Generator(np.random.PCG64(4)).random((4, 2, 2)) * 0.05)
    clu = np.asarray([0, 1, 1, 0, 1, 0, 0, 1, 1], dtype=np.uint16)
    src = tmp_path / "src.npz"
    np.savez(src, names=np.asarray(names), kmers=kmers, sketches=sk, tbl=tbl, clu=clu)
    prefix = str(tmp_path / "db" / "db")
    script = r'''
import sys, os
import numpy as np, h5py
src, prefix = sys.argv[1:3]
z = np.load(src)
os.makedirs(os.path.dirname(prefix))
with h5py.File(prefix + ".h5", "w") as f:
    g = f.create_group("sketches")
    g.attrs["sketch_version"] = "test"
    g.attrs["codon_phased"] = False
    for i, nm in enumerate(z["names"]):
        s = g.create_group(str(nm))
        s.attrs["sketchsize64"] = 3
        s.attrs["bbits"] = 14
        s.attrs["length"] = 2000000 + i
        s.attrs["missing_bases"] = 0
        s.attrs["base_freq"] = [0.25, 0.25, 0.3, 0.2]
        s.attrs["kmers"] = [int(k) for k in z["kmers"]]
        for j, k in enumerate(z["kmers"]):
            d = s.create_dataset(str(int(k)), data=z["sketches"][i, j], dtype="uint64")
            d.attrs["kmer-size"] = int(k)
    r = f.create_group("random")
    r.attrs["k_min"] = np.uint32(13); r.attrs["k_max"] = np.uint32(25); r.attrs["use_rc"] = True
    r.create_dataset("table_keys", data=np.asarray([str(n).encode() for n in z["names"]]))
    r.create_dataset("table_values", data=z["clu"].astype("uint16"))
    r.create_dataset("matches_keys", data=z["kmers"].astype("uint64"))
    r.create_dataset("matches_values", data=z["tbl"].reshape(4, 4))
    r.create_dataset("centroids", data=np.asarray([[0.25, 0.25, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2]]))
'''
    r = subprocess.run([H5_PYTHON, "-c", script, str(src), prefix], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # native read: subset / order of names and k, attributes, the [EXT]-mapped random table
    got = sketchdb.load(prefix, ["s07", "s02", "s05"], [21, 13])
    assert got.sketchsize64 == 3 and got.bbits == 14 and list(got.kmers) == [21, 13]
    assert np.array_equal(got.sketches, sk[[7, 2, 5]][:, [2, 0], :])
    assert got.random_status == "mapped" and list(got.clusters) == [1, 1, 0]
    assert np.array_equal(got.random_table, tbl[[2, 0]].astype(np.float32))
    assert list(got.lengths) == [2000007, 2000002, 2000005] and np.allclose(got.base_freq[0], [0.25, 0.25, 0.3, 0.2])
    assert sketchdb.getSeqsInDb(prefix + ".h5") == names
    ks, s64, phased = sketchdb.readDBParams(os.path.dirname(prefix))
    assert list(ks) == [13, 17, 21, 25] and s64 == 3 and phased is False
    # conversion carries everything, the /random group verbatim
    assert sketchdb.convert_h5_to_npz(prefix, prefix + "_conv") == (9, [13, 17, 21, 25], "mapped")
    conv = sketchdb.load(prefix + "_conv", names, kmers)
    assert np.array_equal(conv.sketches, sk) and conv.sketchsize64 == 3 and conv.bbits == 14
    assert np.array_equal(conv.random_table, tbl.astype(np.float32)) and np.array_equal(conv.clusters, clu)
    assert sorted(conv.random_raw) == ["@k_max", "@k_min", "@use_rc", "centroids", "matches_keys",
                                       "matches_values", "table_keys", "table_values"]
    assert np.array_equal(conv.random_raw["matches_values"], tbl.reshape(4, 4))
    # a /random group in some other layout is carried, reported, and NOT guessed at
    script2 = r'''
import sys, h5py, numpy as np
with h5py.File(sys.argv[1], "r+") as f:
    del f["random"]
    r = f.create_group("random"); r.create_dataset("something_else", data=np.arange(5)); r.attrs["v"] = 3
'''
    assert subprocess.run([H5_PYTHON, "-c", script2, prefix + ".h5"], capture_output=True).returncode == 0
    odd = sketchdb.load(prefix, names, kmers)
    assert odd.random_status == "unrecognised" and odd.random_table is None
    assert np.array_equal(odd.random_raw["something_else"], np.arange(5)) and int(odd.random_raw["@v"]) == 3
    # writer -> h5py reads it back (layout of web.py:14-61 + the raw /random group)
    out = str(tmp_path / "w" / "w")
    sketchdb.save_h5(out, names, kmers, sk, 3, 14, random_raw=conv.random_raw, lengths=conv.lengths,
                     base_freq=conv.base_freq, sketch_version="test")
    script3 = r'''
import sys, h5py, numpy as np
z = np.load(sys.argv[2])
with h5py.File(sys.argv[1], "r") as f:
    g = f["sketches"]
    assert g.attrs["sketch_version"] == "test" and not g.attrs["codon_phased"]
    assert sorted(g.keys()) == sorted(str(n) for n in z["names"])
    for i, nm in enumerate(z["names"]):
        s = g[str(nm)]
        assert int(s.attrs["sketchsize64"]) == 3 and int(s.attrs["bbits"]) == 14 and int(s.attrs["length"]) == 2000000 + i
        assert list(s.attrs["kmers"]) == [int(k) for k in z["kmers"]]
        for j, k in enumerate(z["kmers"]):
            assert s[str(int(k))].dtype == np.uint64 and np.array_equal(s[str(int(k))][:], z["sketches"][i, j])
            assert int(s[str(int(k))].attrs["kmer-size"]) == int(k)
    r = f["random"]
    assert int(r.attrs["k_min"]) == 13 and int(r.attrs["k_max"]) == 25
    assert [x.decode() for x in r["table_keys"][:]] == [str(n) for n in z["names"]]
    assert np.array_equal(r["matches_values"][:], z["tbl"].reshape(4, 4)) and np.array_equal(r["table_values"][:], z["clu"])
print("ok")
'''
    r = subprocess.run([H5_PYTHON, "-c", script3, out + ".h5", str(src)], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    back = sketchdb.load(out, names, kmers)
    assert np.array_equal(back.sketches, sk) and np.array_equal(back.random_table, tbl.astype(np.float32))


def test_random_correct_on_a_database_without_a_table_raises(tmp_path, monkeypatch):
    """PopPUNK never queries a database without random match chances (PopPUNK/sketchlib.py:455-466);
    random_correct=True on one must not silently return uncorrected distances."""
    prefix, names, _ = make_db(tmp_path, "norand", 6, with_random=False)
    db = prefix + "/norand"
    monkeypatch.delenv("PPK_ALLOW_NO_RANDOM", raising=False)
    pp_sketchlib._DB_CACHE.clear()
    with pytest.raises(RuntimeError, match="no random match chances"):
        pp_sketchlib.queryDatabase(db, db, names, names, [13, 17, 21], True, False, 1, False, 0)
    with pytest.raises(RuntimeError, match="no random match chances"):
        sketchlib.queryDatabase(names, names, prefix, prefix, [13, 17, 21])


def test_version_is_parsed_by_the_reference_check():
    """checkSketchlibVersion does [int(v) for v in version.split('.')] (PopPUNK/sketchlib.py:49-50)
    and wants >= 2.0.1 (PopPUNK/__init__.py:9-11)."""
    v = [int(x) for x in pp_sketchlib.version.split(".")]
    assert len(v) == 3 and tuple(v) >= (2, 0, 1)


def test_fitKmerCurve_mirror_against_the_reference_goldens(golden_dir):
    """tests/golden/fit_kmer_curve.json (the reference's fitKmerCurve run by make_golden.py): the
    mirror solves the same bounded least-squares problem exactly, INCLUDING the cases where a
    bound is active (which the regression of the distance kernel clamps instead)."""
    import json
    g = json.load(open(os.path.join(golden_dir, "fit_kmer_curve.json")))
    n_active = 0
    for c in g["cases"]:
        got = sketchlib.fitKmerCurve(np.asarray(c["jaccard"]), np.asarray(c["klist"]))
        tol = 5e-6 if c["interior"] else 1e-4          # the reference's trust-region solver stops early on a bound
        assert abs(got[0] - c["core"]) <= tol and abs(got[1] - c["accessory"]) <= tol, c
        # *_tight: the same reference function with only scipy's stopping tolerances at 1e-15 (make_golden.py):
        # the exact bounded minimiser, bound-active cases included
        assert abs(got[0] - c["core_tight"]) <= 1e-8 and abs(got[1] - c["accessory_tight"]) <= 1e-8, c
        n_active += not c["interior"]
    assert n_active >= 5
    assert list(sketchlib.fitKmerCurve(np.asarray([0.5, 0.0]), np.asarray([13, 17]))) == [0, 0]


def test_sub_sample_requery_is_sliced_from_the_loaded_database(tmp_path, monkeypatch):
    """The --plot-fit leg re-queries single samples of the database it has just queried
    (PopPUNK/sketchlib.py:547-564).  Those requests are sliced from the loaded full database: the file
    is not read again and the full database (with its resident copies) is not pushed out of the
    four-entry cache (round-2 advisor finding)."""
    prefix, names, sk = make_db(tmp_path, "full", 12)
    db = prefix + "/full"
    pp_sketchlib._DB_CACHE.clear()
    loads = []
    real = sketchdb.load
    monkeypatch.setattr(sketchdb, "load", lambda *a, **k: (loads.append(a[1]), real(*a, **k))[1])
    full = pp_sketchlib._load_cached(db, names, [13, 17, 21])
    assert len(loads) == 1 and not full.transient and len(pp_sketchlib._DB_CACHE) == 1
    for pick in ([names[3]], [names[7]], [names[11], names[0]], [names[5]], [names[6]], [names[2]]):
        e = pp_sketchlib._load_cached(db, pick, [13, 17, 21])
        assert e.transient and e.loaded.names == pick
        rows = [names.index(p) for p in pick]
        assert np.array_equal(e.loaded.sketches, sk[rows]) and e.loaded.sketches.flags["C_CONTIGUOUS"]
        assert e.loaded.random_table is full.loaded.random_table
        assert np.array_equal(e.loaded.clusters, full.loaded.clusters[rows])
    assert len(loads) == 1 and list(pp_sketchlib._DB_CACHE.values()) == [full]
    assert pp_sketchlib._load_cached(db, names, [13, 17, 21]) is full
    # another k list is another load; a name the database lacks still fails in the loader
    other = pp_sketchlib._load_cached(db, names, [13, 21])
    assert len(loads) == 2 and other is not full and other.loaded.sketches.shape[1] == 2
    with pytest.raises(RuntimeError, match="not found"):
        pp_sketchlib._load_cached(db, ["nobody"], [13, 17, 21])
    pp_sketchlib._DB_CACHE.clear()


def test_stderr_redirected_is_fd_level(tmp_path, capfd):
    """PopPUNK wraps the --plot-fit re-queries in an fd-level redirect (PopPUNK/utils.py:61-83,
    PopPUNK/sketchlib.py:546): native write(2, ...) output is silenced too, and fd 2 works again after."""
    from poppunk_amd.utils import stderr_redirected
    log = str(tmp_path / "err.txt")
    os.write(2, b"before\n")
    with stderr_redirected(to=log):
        os.write(2, b"native meter\n")
        sys.stderr.write("python side\n")
        sys.stderr.flush()
    os.write(2, b"after\n")
    err = capfd.readouterr().err
    assert "before" in err and "after" in err and "native" not in err and "python side" not in err
    got = open(log).read()
    assert "native meter" in got and "python side" in got
    with pytest.raises(ValueError):
        with stderr_redirected():
            raise ValueError("restored on exceptions too")
    os.write(2, b"still there\n")
    assert "still there" in capfd.readouterr().err


def _route(n_ref, nk, s64, q_rows=None, self_=1, mask=0, knn=0, geometry=(256, 8, 512), bbits=14,
           knobs=(1200, 215, 1, 1, 0, 0, 2048 << 20)):
    import ctypes as C
    from poppunk_amd import _lib
    kn = (C.c_longlong * 7)(*knobs)
    route, slices = C.c_int(-1), C.c_int(-1)
    tiles, limit = C.c_size_t(0), C.c_size_t(0)
    rc = _lib.lib().ppk_choose_route(n_ref, n_ref if q_rows is None else q_rows, self_, nk, s64, bbits, mask, knn,
                                     geometry[0], geometry[1], geometry[2], kn, C.byref(route), C.byref(slices),
                                     C.byref(tiles), C.byref(limit))
    assert rc == 0
    return route.value, slices.value, tiles.value, limit.value


def test_route_choice_is_a_pure_function_pinned_on_mi355x_and_sane_elsewhere():
    """ppk_choose_route (ppk_dist.hip): the kernel shape a band runs through, from the job's shape, the device's
    geometry (read from the device, never assumed) and the options.  Pinned: today's choices on MI355X in SPX mode
    (256 CUs, 8 XCDs, 512 resident tile workgroups) for shapes measured in profiles/r05/ksplit_*.txt; and a
    CPX-shaped device (32 CUs, one XCD, 64 slots) scales every tile-count rule by its slots.  No GPU is touched."""
    TILE, ONE, TWO, WIDE, UNFUSED = 0, 1, 2, 3, 4
    # s = 1 024, the default 5 k: tiles fitted from the LDS table, the one-launch form up to 1 200 tiles
    assert _route(1000, 5, 16)[0] == ONE and _route(1000, 5, 16)[2] == 96
    assert _route(4000, 5, 16)[0] == ONE and _route(4000, 5, 16)[2] == 1125
    assert _route(10000, 5, 16) == (TILE, 1, 6573, 1200)
    # tiny jobs cut every k into pieces while that stays within one round of slots
    assert _route(300, 5, 16)[1] == 4 and _route(600, 5, 16)[1] == 2 and _route(1000, 5, 16)[1] == 1
    # s = 1 024 with k lists the LDS table does not serve: level at 700 tiles whatever nk (ksplit_s1024_other_shapes)
    assert _route(3000, 9, 16)[0] == ONE and _route(4000, 9, 16)[0] == TILE
    assert _route(3000, 10, 16)[0] == ONE and _route(4000, 10, 16)[0] == TILE
    assert _route(2000, 17, 16)[0] == ONE and _route(4000, 17, 16)[0] == WIDE      # 17 x 11 bits > 128
    # PopPUNK's default sketch size: k-split at any size its scratch allows (ksplit_long_sketches)
    for n in (600, 3000, 10000, 20000):
        assert _route(n, 5, 156)[0] == ONE, n
    assert _route(10000, 10, 156)[0] == ONE            # the bench's wide_k leg: fit from parts
    assert _route(30000, 5, 156)[0] == TILE            # 58 000 tiles x 5 x 16 KB: beyond the 2 GB the path may take
    assert _route(30000, 10, 156)[0] == WIDE
    assert _route(3000, 5, 156, knobs=(1200, 215, 0, 1, 0, 0, 2048 << 20))[0] == ONE       # ksplit_long 0: 658 <= 700 tiles
    assert _route(4000, 5, 156, knobs=(1200, 215, 0, 1, 0, 0, 2048 << 20))[0] == TILE
    assert _route(10000, 5, 156, knobs=(1200, 215, 1, 1, 0, 0, 64 << 20))[0] == TILE       # a 64 MB allowance
    # switches
    assert _route(1000, 5, 16, knobs=(0, 215, 1, 1, 0, 0, 2048 << 20))[0] == TILE          # ksplit 0
    assert _route(1000, 5, 16, knobs=(1200, 215, 1, 0, 0, 0, 2048 << 20))[0] == TWO        # ksplit_fused 0
    assert _route(1000, 5, 16, knobs=(1200, 215, 1, 1, 2, 0, 2048 << 20))[1] == 2          # ksplit_slices 2
    assert _route(1000, 5, 16, knobs=(1200, 215, 1, 1, 0, 2, 2048 << 20))[0] == ONE        # wide_kpg 2: from the parts
    assert _route(10000, 5, 16, knobs=(1200, 215, 1, 1, 0, 2, 2048 << 20))[0] == WIDE
    # modes: neighbours always the tile kernel; the fused boundary mode k-splits in its one-launch form only
    assert _route(1000, 5, 16, knn=1)[0] == TILE
    assert _route(1000, 5, 16, mask=1)[0] == ONE
    assert _route(1000, 5, 16, mask=1, knobs=(1200, 215, 1, 0, 0, 0, 2048 << 20))[0] == TILE
    assert _route(1000, 5, 1, mask=1)[0] == TILE                                           # one-block sketches
    # ref x query: poppunk_assign's few queries against many refs
    assert _route(10000, 5, 16, q_rows=64, self_=0)[0] == ONE
    assert _route(10000, 5, 16, q_rows=50000, self_=0)[0] == TILE
    # other bbits (never written by PopPUNK): the generic kernel; beyond 128 count bits the unfused counts route
    assert _route(1000, 5, 16, bbits=8)[0] == TILE
    assert _route(1000, 17, 16, bbits=8)[0] == UNFUSED
    # a CPX-shaped partition: an eighth of the slots, an eighth of every tile-count threshold, no XCD padding
    cpx = (32, 1, 64)
    assert _route(1000, 5, 16, geometry=cpx) == (ONE, 1, 96, 150)
    assert _route(2000, 5, 16, geometry=cpx)[0] == TILE
    assert _route(300, 5, 16, geometry=cpx)[1] == 1 and _route(100, 5, 16, geometry=cpx)[1] == 2 and _route(50, 5, 16, geometry=cpx)[1] == 4
    assert _route(1000, 9, 16, geometry=cpx)[3] == 87 and _route(1000, 9, 16, geometry=cpx)[0] == TILE
    assert _route(3000, 5, 156, geometry=cpx)[0] == ONE
    # bad arguments are errors, not routes
    import ctypes as C
    from poppunk_amd import _lib
    assert _lib.lib().ppk_choose_route(10, 10, 1, 5, 16, 14, 0, 0, 0, 8, 512, (C.c_longlong * 7)(), C.byref(C.c_int()),
                                       None, None, None) != 0


def test_sweep_filter_rounding_argument_holds_on_float32():
    """The boundary sweep's classify pass (ppk_iterate.hip, ti1_classify_kernel FILTER) drops a row after ONE evaluation
    when x, y >= 0 and fl(fl(y x_L) + fl(x y_L)) > c_L (1 + 2^-20) for the outermost boundary L, claiming that every
    boundary inside L (x_o <= x_L, y_o <= y_L, all >= 2^-40) then excludes it as well: fl(fl(y x_o) + fl(x y_o)) >
    fl(x_o y_o).  Numpy float32 arithmetic is the same IEEE arithmetic, un-fused: 20 million rows placed within a few
    ulps of the filter's threshold (where a wrong margin would show), across magnitudes from 2^-38 to 2^30, against 16
    nested boundaries each -- no row the filter drops may be within any of them."""
    rng = np.random.Generator(np.random.PCG64(20260929))
    f32 = np.float32
    worst = 0
    for trial in range(40):
        mag = f32(2.0) ** f32(rng.integers(-38, 31))
        xL = f32(mag * f32(rng.uniform(0.5, 2.0)))
        yL = f32(xL * f32(2.0) ** f32(rng.integers(-6, 7)) * f32(rng.uniform(0.5, 2.0)))
        if not (np.isfinite(xL * yL) and xL >= f32(2.0) ** -40 and yL >= f32(2.0) ** -40):
            continue
        cL = f32(xL * yL)
        thr = f32(cL * f32(1.00000095367431640625))
        n = 500000
        # points on the line y / yL + x / xL = 1 + e, e within +-2^-19 of the filter's margin and a few far ones
        t = rng.random(n).astype(np.float64)
        e = np.concatenate([rng.uniform(-2.0 ** -19, 2.0 ** -18, n - n // 10), rng.uniform(-0.5, 2.0, n // 10)])
        x = (t * (1.0 + e) * float(xL)).astype(f32)
        y = ((1.0 - t) * (1.0 + e) * float(yL)).astype(f32)
        x[:50] = 0
        y[50:100] = 0
        aL = (y * xL).astype(f32) + (x * yL).astype(f32)
        dropped = (aL > thr) & (x >= 0) & (y >= 0)
        worst = max(worst, int(dropped.sum()))
        for _ in range(16):
            xo = f32(max(float(xL) * rng.uniform(2.0 ** -12, 1.0), 2.0 ** -40))
            yo = f32(max(float(yL) * rng.uniform(2.0 ** -12, 1.0), 2.0 ** -40))
            assert xo <= xL and yo <= yL
            ao = (y * xo).astype(f32) + (x * yo).astype(f32)
            within = ao <= f32(xo * yo)
            assert not np.any(within & dropped), (trial, float(xL), float(yL), float(xo), float(yo))
        # the same boundary itself
        assert not np.any((aL <= cL) & dropped)
    assert worst > 100000      # (the filter did drop rows in these trials: the check is not vacuous)


def test_sweep_bisection_rounding_arguments_hold_on_float32():
    """The bisection of the sweeps' classify pass (ppk_iterate.hip, probe_rows_slope2) rests on two claims about
    a_o = fl(fl(y x_o) + fl(x y_o)) for x, y >= 0 and boundaries nested outwards (x_o, y_o non-decreasing, >= 2^-40):
      a_o < fl(c_o (1 - 2^-20))  =>  the row is WITHIN every boundary o' around o:  a_o' <= c_o' = fl(x_o' y_o');
      a_o > fl(c_o (1 + 2^-20))  =>  the row is OUTSIDE every boundary o' inside o:  a_o' >  c_o'
    (the second is the filter's argument with L = o).  Numpy float32 is the same un-fused IEEE arithmetic: rows placed
    within a few 2^-20 of boundary o, either side, across magnitudes, against 16 boundaries around it and 16 inside it."""
    rng = np.random.Generator(np.random.PCG64(20260930))
    f32 = np.float32
    n_in = n_out = 0
    for trial in range(40):
        mag = f32(2.0) ** f32(rng.integers(-30, 25))
        xo = f32(mag * f32(rng.uniform(0.5, 2.0)))
        yo = f32(xo * f32(2.0) ** f32(rng.integers(-6, 7)) * f32(rng.uniform(0.5, 2.0)))
        if not (np.isfinite(xo * yo) and xo >= f32(2.0) ** -40 and yo >= f32(2.0) ** -40):
            continue
        co = f32(xo * yo)
        lo, hi = f32(co * f32(0.99999904632568359375)), f32(co * f32(1.00000095367431640625))
        n = 400000
        t = rng.random(n).astype(np.float64)
        e = np.concatenate([rng.uniform(-2.0 ** -18, 2.0 ** -18, n - n // 10), rng.uniform(-0.9, 2.0, n // 10)])
        x = (t * (1.0 + e) * float(xo)).astype(f32)
        y = ((1.0 - t) * (1.0 + e) * float(yo)).astype(f32)
        x[:50] = 0
        y[50:100] = 0
        a = (y * xo).astype(f32) + (x * yo).astype(f32)
        safely_in, safely_out = a < lo, a > hi
        n_in += int(safely_in.sum())
        n_out += int(safely_out.sum())
        # the boundary itself
        assert np.all(a[safely_in] <= co) and np.all(a[safely_out] > co)
        for _ in range(16):      # boundaries AROUND o (also barely around it: factors down to 1 + 2^-23)
            fx = 1.0 + 2.0 ** rng.uniform(-23, 3)
            fy = 1.0 + 2.0 ** rng.uniform(-23, 3)
            x2, y2 = f32(float(xo) * fx), f32(float(yo) * fy)
            if not (x2 >= xo and y2 >= yo and np.isfinite(x2 * y2)):
                continue
            a2 = (y * x2).astype(f32) + (x * y2).astype(f32)
            assert np.all(a2[safely_in] <= f32(x2 * y2)), (trial, float(xo), float(yo), float(x2), float(y2))
        for _ in range(16):      # boundaries INSIDE o
            x1 = f32(max(float(xo) * (1.0 - 2.0 ** rng.uniform(-23, -0.01)), 2.0 ** -40))
            y1 = f32(max(float(yo) * (1.0 - 2.0 ** rng.uniform(-23, -0.01)), 2.0 ** -40))
            if not (x1 <= xo and y1 <= yo):
                continue
            a1 = (y * x1).astype(f32) + (x * y1).astype(f32)
            assert np.all(a1[safely_out] > f32(x1 * y1)), (trial, float(xo), float(yo), float(x1), float(y1))
    assert n_in > 1000000 and n_out > 1000000      # (not vacuous)
    # ... and the margin is what carries them: with none (a_o <= c_o / a_o > c_o taken as "safely"), rounding does flip
    # verdicts between boundaries that lie within 2^-18 of one another
    flips = 0
    rng2 = np.random.Generator(np.random.PCG64(1))
    for trial in range(40):
        mag = f32(2.0) ** f32(rng2.integers(-30, 25))
        xo = f32(mag * f32(rng2.uniform(0.5, 2.0)))
        yo = f32(xo * f32(2.0) ** f32(rng2.integers(-6, 7)) * f32(rng2.uniform(0.5, 2.0)))
        co = f32(xo * yo)
        n = 400000
        t = rng2.random(n).astype(np.float64)
        e = rng2.uniform(-2.0 ** -22, 2.0 ** -22, n)
        x = (t * (1.0 + e) * float(xo)).astype(f32)
        y = ((1.0 - t) * (1.0 + e) * float(yo)).astype(f32)
        a = (y * xo).astype(f32) + (x * yo).astype(f32)
        for _ in range(16):
            x2 = f32(float(xo) * (1.0 + 2.0 ** rng2.uniform(-23, -18)))
            y2 = f32(float(yo) * (1.0 + 2.0 ** rng2.uniform(-23, -18)))
            a2 = (y * x2).astype(f32) + (x * y2).astype(f32)
            flips += int(np.sum(a2[a <= co] > f32(x2 * y2)))
            x1 = f32(float(xo) * (1.0 - 2.0 ** rng2.uniform(-23, -18)))
            y1 = f32(float(yo) * (1.0 - 2.0 ** rng2.uniform(-23, -18)))
            a1 = (y * x1).astype(f32) + (x * y1).astype(f32)
            flips += int(np.sum(a1[a > co] <= f32(x1 * y1)))
    assert flips > 0


def _sweep_plan(x_max, y_max, slope=2, one_d=True):
    import ctypes as C
    from poppunk_amd import _lib
    xm = np.ascontiguousarray(x_max, dtype=np.float32)
    ym = np.ascontiguousarray(np.broadcast_to(np.asarray(y_max, dtype=np.float32), xm.shape))
    out = (C.c_int * 4)()
    rc = _lib.lib().ppk_sweep_plan(xm.ctypes.data_as(C.POINTER(C.c_float)), ym.ctypes.data_as(C.POINTER(C.c_float)), xm.size,
                                   slope, 1 if one_d else 0, out)
    assert rc == 0
    return tuple(out)      # (mode, filter, window, guess)


def test_sweep_plan_is_a_pure_function_of_the_boundaries():
    """ppk_sweep_plan (ppk_iterate.hip): which form of the sweeps' classify pass a list of boundaries takes -- the early
    exit (one boundary contains the rest), the bisection (nested outwards in order), the guessed end indices (parallel,
    evenly spaced; the 1-D sweep only) -- decided on the host from the boundaries alone."""
    g = 1.12
    xm = np.linspace(0.117, 0.9, 40)
    # refine's outward sweep over a linspace of offsets: everything on
    assert _sweep_plan(xm, xm / g) == (2, 1, 1, 1)
    # the 2-D sweep (one y_max, x_max ascending): nested, not parallel, and no keys -> no guess
    assert _sweep_plan(xm, 0.3, one_d=False) == (2, 1, 1, 0)
    assert _sweep_plan(xm, 0.3, one_d=True) == (2, 1, 1, 0)
    # unevenly spaced offsets: bisection without the guess
    un = np.sort(np.concatenate([xm[:20], xm[20:] ** 1.3]))
    assert _sweep_plan(un, un / g) == (2, 1, 1, 0)
    # an inward sweep: the first boundary contains the rest (filter), the order is not outwards (no bisection)
    assert _sweep_plan(xm[::-1], xm[::-1] / g) == (2, 1, 0, 0)
    # boundaries that cross (x_max grows, y_max shrinks): no boundary contains the others
    assert _sweep_plan(xm, (xm / g)[::-1]) == (2, 0, 0, 0)
    # a boundary on an axis, a tiny one, an infinite one: the reference's line_dist as it stands / no margin argument
    assert _sweep_plan([0.0, 0.2, 0.3], [0.1, 0.2, 0.3])[0] == 3
    assert _sweep_plan([1e-13, 0.2, 0.3], [1e-13, 0.2, 0.3]) == (2, 0, 0, 0)
    assert _sweep_plan([0.1, 0.2, np.inf], [0.1, 0.2, 0.3])[0] == 3
    # slopes 0 and 1 compare one coordinate: none of this applies
    assert _sweep_plan(xm, xm / g, slope=0) == (0, 0, 0, 0)
    assert _sweep_plan(xm, xm / g, slope=1) == (1, 0, 0, 0)
    # two boundaries: too few for a spacing
    assert _sweep_plan(xm[:2], xm[:2] / g) == (2, 1, 1, 0)
