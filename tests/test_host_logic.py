"""CPU tests of the host-side mirror of the reference interface: argument handling and error
behaviour of sketchlib.queryDatabase (PopPUNK/sketchlib.py:475-632), the noconvert rule of
poppunk_refine (src/python_bindings.cpp:82,:89), the sketch database files and the synthetic
generator.  No GPU compute is triggered here."""
import os

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
def test_reference_h5_layout_reader_and_converter(tmp_path):
    """The reference keeps sketches in <prefix>/<basename>.h5: group /sketches, one group per sample
    with attrs sketchsize64 / bbits / kmers / length / ..., one uint64 dataset per k named str(k)
    (PopPUNK/web.py:14-61).  h5py lives in the image's conda interpreter only, so that interpreter
    writes such a file, reads it through sketchdb.load (subset and order of names and of k) and
    converts it; this interpreter then reads the converted .npz and compares everything."""
    import subprocess
    from poppunk_amd import sketchdb, synth
    kmers = np.asarray([13, 17, 21, 25], dtype=np.int32)
    sk, _ = synth.make_sketches(9, kmers, sketchsize64=3, bbits=14, cluster_size=3, seed=12)
    names = ["s%02d" % i for i in range(9)]
    src = tmp_path / "src.npz"
    np.savez(src, names=np.asarray(names), kmers=kmers, sketches=sk)
    prefix = str(tmp_path / "db" / "db")
    script = r'''
import sys, os, importlib.util
import numpy as np, h5py
src, prefix, repo, out = sys.argv[1:5]
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
        s.attrs["length"] = 2000000
        s.attrs["missing_bases"] = 0
        s.attrs["base_freq"] = [0.25, 0.25, 0.25, 0.25]
        s.attrs["kmers"] = [int(k) for k in z["kmers"]]
        for j, k in enumerate(z["kmers"]):
            d = s.create_dataset(str(int(k)), data=z["sketches"][i, j], dtype="uint64")
            d.attrs["kmer-size"] = int(k)
spec = importlib.util.spec_from_file_location("sketchdb", os.path.join(repo, "poppunk_amd", "sketchdb.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
got = m.load(prefix, ["s07", "s02", "s05"], [21, 13])
np.savez(out, sketches=got.sketches, kmers=got.kmers, s64=got.sketchsize64, bbits=got.bbits)
print(m.convert_h5_to_npz(prefix, prefix + "_conv"))
'''
    out = tmp_path / "loaded.npz"
    r = subprocess.run([H5_PYTHON, "-c", script, str(src), prefix, ROOT, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    assert int(z["s64"]) == 3 and int(z["bbits"]) == 14 and list(z["kmers"]) == [21, 13]
    assert np.array_equal(z["sketches"], sk[[7, 2, 5]][:, [2, 0], :])
    conv = sketchdb.load(prefix + "_conv", names, kmers)              # .npz written by the converter
    assert np.array_equal(conv.sketches, sk) and conv.sketchsize64 == 3 and conv.bbits == 14
    # without h5py (this interpreter) a .h5 database says how to convert it
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="poppunk_amd.sketchdb"):
            sketchdb.load(prefix, names, kmers)
