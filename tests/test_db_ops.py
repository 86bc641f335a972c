"""The sketch-database operations either side of the distance call -- joinDBs, removeFromDB and the
parameter readers (PopPUNK/sketchlib.py:109-346,:672-690) -- against files and return values the
REFERENCE's own functions produced from the same two input databases (tests/golden/make_golden_db.py,
run once under h5py; inputs tests/golden/db/*.h5, expected content tests/golden/db_ops.json)."""
import json
import os
import shutil

import numpy as np
import pytest

from poppunk_amd import h5lite, sketchdb, sketchlib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

pytestmark = pytest.mark.skipif(not h5lite.available(), reason="libhdf5 not found")


def _plain(v):
    a = np.asarray(v)
    if a.dtype.kind in "SUO":
        return {"kind": "str", "shape": list(a.shape),
                "values": [x.decode() if isinstance(x, bytes) else str(x) for x in a.ravel().tolist()]}
    kind = "b" if a.dtype.kind == "b" else a.dtype.kind
    return {"kind": kind, "shape": list(a.shape),
            "values": [int(x) if kind in "iub" else float(x) for x in a.ravel().tolist()]}


def listing(path):
    out = {}

    def visit(group, prefix):
        for nm in group.keys():
            obj = group[nm]
            p = prefix + "/" + nm
            is_ds = isinstance(obj, h5lite.Dataset)
            e = {"type": "dataset" if is_ds else "group", "attrs": {k: _plain(v) for k, v in sorted(obj.attrs.items())}}
            if is_ds:
                e["data"] = _plain(obj.read())
            out[p] = e
            if not is_ds:
                visit(obj, p)

    with h5lite.File(path, "r") as f:
        visit(f, "")
    return out


def _same(got, want):
    assert sorted(got) == sorted(want)
    for path in want:
        g, w = got[path], want[path]
        assert g["type"] == w["type"], path
        assert sorted(g["attrs"]) == sorted(w["attrs"]), path
        for k in w["attrs"]:
            _same_value(g["attrs"][k], w["attrs"][k], path + "@" + k)
        if w["type"] == "dataset":
            _same_value(g["data"], w["data"], path)


def _same_value(g, w, what):
    # h5py hands an enum-over-int8 attribute (its bool) back as bool, h5lite as the stored integers
    gk, wk = ("i" if g["kind"] in "bu" else g["kind"]), ("i" if w["kind"] in "bu" else w["kind"])
    assert gk == wk, what
    assert g["shape"] == w["shape"], what
    assert g["values"] == w["values"], what        # floats too: copied bytes, not recomputed numbers


@pytest.fixture()
def dbs(tmp_path):
    for nm in ("a", "b"):
        os.makedirs(tmp_path / nm)
        shutil.copy(os.path.join(GOLD, "db", nm + ".h5"), tmp_path / nm / (nm + ".h5"))
    return tmp_path


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "db_ops.json")) as f:
        return json.load(f)


def test_reader_sees_the_inputs_as_h5py_does(dbs, gold):
    _same(listing(str(dbs / "a" / "a.h5")), gold["inputs"]["a"])
    _same(listing(str(dbs / "b" / "b.h5")), gold["inputs"]["b"])


def test_joinDBs(dbs, gold, capfd):
    os.makedirs(dbs / "joined")
    sketchlib.joinDBs(str(dbs / "a"), str(dbs / "b"), str(dbs / "joined"))
    _same(listing(str(dbs / "joined" / "joined.h5")), gold["join_ab"])
    assert not os.path.exists(dbs / "joined" / "joined.tmp.h5")
    assert sketchlib.getSeqsInDb(str(dbs / "joined" / "joined.h5")) == gold["getSeqsInDb_joined"]
    sketchlib.joinDBs(str(dbs / "b" / "b.h5"), str(dbs / "a" / "a.h5"), str(dbs / "ba"), full_names=True)
    _same(listing(str(dbs / "ba.h5")), gold["join_ba_full"])
    # the output may be one of the inputs (PopPUNK/assign.py:741 joins into `output` itself)
    sketchlib.joinDBs(str(dbs / "a"), str(dbs / "b"), str(dbs / "b"))
    _same(listing(str(dbs / "b" / "b.h5")), gold["join_ab"])
    capfd.readouterr()
    # update_random: the table of db1 is carried, and the call says so (addRandom is sketching-side)
    os.makedirs(dbs / "j2")
    shutil.copy(os.path.join(GOLD, "db", "b.h5"), dbs / "b" / "b.h5")
    sketchlib.joinDBs(str(dbs / "a"), str(dbs / "b"), str(dbs / "j2"), update_random={"threads": 2})
    assert "carried over" in capfd.readouterr().err
    _same(listing(str(dbs / "j2" / "j2.h5")), gold["join_ab"])


def test_joined_database_is_queryable_and_names_clash_is_an_error(dbs):
    os.makedirs(dbs / "joined")
    sketchlib.joinDBs(str(dbs / "a"), str(dbs / "b"), str(dbs / "joined"))
    names = sketchlib.getSeqsInDb(str(dbs / "joined" / "joined.h5"))
    got = sketchdb.load(str(dbs / "joined" / "joined"), names, [13, 17, 21])
    a = sketchdb.load(str(dbs / "a" / "a"), ["s_gamma"], [13, 17, 21])
    assert got.sketches.shape == (8, 3, 6)
    assert np.array_equal(got.sketches[names.index("s_gamma")], a.sketches[0])
    # a sample that is in both files: the library refuses the second copy, the mirror exits like the reference
    with pytest.raises(SystemExit):
        sketchlib.joinDBs(str(dbs / "a"), str(dbs / "a"), str(dbs / "joined"))


def test_removeFromDB(dbs, gold, capfd):
    os.makedirs(dbs / "pruned")
    sketchlib.removeFromDB(str(dbs / "a"), str(dbs / "pruned"), ["s_beta", "s_eps", "not_there"])
    err = capfd.readouterr().err
    assert "WARNING: Did not find samples to remove:" in err and "not_there" in err
    _same(listing(str(dbs / "pruned" / "pruned.tmp.h5")), gold["remove_a"])
    sketchlib.removeFromDB(str(dbs / "b" / "b.h5"), str(dbs / "b_same.h5"), [], full_names=True)
    _same(listing(str(dbs / "b_same.h5")), gold["remove_none_b_full"])
    assert capfd.readouterr().err == ""


def test_parameter_readers(dbs, gold):
    kmers, s, cp = sketchlib.readDBParams(str(dbs / "a"))
    want = gold["readDBParams_a"]
    assert [int(k) for k in kmers] == want["kmers"] and int(s) == want["sketch_size"] and bool(cp) == want["codon_phased"]
    assert [int(k) for k in sketchlib.getKmersFromReferenceDatabase(str(dbs / "a"))] == want["kmers"]
    assert sketchlib.getSketchSize(str(dbs / "a")) == (want["sketch_size"], want["codon_phased"])
    assert sketchlib.getSeqsInDb(str(dbs / "a" / "a.h5")) == gold["getSeqsInDb_a"]
    gl, amb = sketchlib.get_database_statistics(str(dbs / "a"))
    assert [int(x) for x in gl] == gold["get_database_statistics_a"]["genome_lengths"]
    assert [int(x) for x in amb] == gold["get_database_statistics_a"]["ambiguous_bases"]
