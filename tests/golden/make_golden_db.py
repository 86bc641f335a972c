#!/opt/conda/bin/python3.9
"""Golden fixtures for the sketch-database operations either side of the distance path
(`--update-db`, QC pruning, reference picking): PopPUNK.sketchlib.joinDBs / removeFromDB /
getSketchSize / getKmersFromReferenceDatabase / readDBParams / getSeqsInDb /
get_database_statistics (PopPUNK/sketchlib.py:109-346, :672-690).

Run in the BUILD container only, under the interpreter that has h5py
(`/opt/conda/bin/python3.9 tests/golden/make_golden_db.py`): the named function definitions are
pulled out of the reference module with `ast` and executed under the real h5py / numpy -- the
module cannot be imported whole (it imports pp_sketchlib) and nothing is stubbed.

Fixtures written (data only):
  db/a.h5, db/b.h5     two small databases in the layout of PopPUNK/web.py:14-61, written here with
                       h5py (a: 5 samples + a /random group, b: 3 samples, one sample name shared
                       with nothing; attributes `length`, `missing_bases`, `base_freq` filled)
  db_ops.json          for every operation: its arguments and a content listing of the file it produced
                       (every group, dataset and attribute: dtype kind, shape, values), plus the
                       return values of the read-only functions
"""
import ast
import json
import os
import shutil
import sys
import tempfile

import h5py
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["joinDBs", "removeFromDB", "getSketchSize", "getKmersFromReferenceDatabase", "readDBParams",
         "getSeqsInDb", "get_database_statistics"]


def extract_functions(path, names, namespace):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
    missing = [n for n in names if n not in namespace]
    if missing:
        raise RuntimeError("not found in %s: %s" % (path, missing))
    return namespace


def listing(path):
    """Every object of an HDF5 file as plain data (the test builds the same listing with h5lite)."""
    out = {}

    def plain(v):
        a = np.asarray(v)
        if a.dtype.kind in "SUO":
            vals = [x.decode() if isinstance(x, bytes) else str(x) for x in a.ravel().tolist()]
            return {"kind": "str", "shape": list(a.shape), "values": vals}
        kind = "b" if a.dtype.kind == "b" else a.dtype.kind
        return {"kind": kind, "shape": list(a.shape),
                "values": [int(x) if kind in "iub" else float(x) for x in a.ravel().tolist()]}

    def visit(name, obj):
        entry = {"type": "dataset" if isinstance(obj, h5py.Dataset) else "group",
                 "attrs": {k: plain(v) for k, v in sorted(obj.attrs.items())}}
        if isinstance(obj, h5py.Dataset):
            entry["data"] = plain(obj[()])
        out["/" + name] = entry

    with h5py.File(path, "r") as f:
        f.visititems(visit)
    return out


def write_db(path, names, seed, with_random):
    rng = np.random.Generator(np.random.PCG64(seed))
    kmers = [13, 17, 21]
    s64, bbits = 2, 3
    with h5py.File(path, "w") as f:
        top = f.create_group("sketches")
        top.attrs["sketch_version"] = "golden-1"
        top.attrs["codon_phased"] = False
        for nm in names:
            g = top.create_group(nm)
            g.attrs["sketchsize64"] = s64
            g.attrs["bbits"] = bbits
            g.attrs["kmers"] = kmers
            g.attrs["length"] = int(rng.integers(1_900_000, 2_300_000))
            g.attrs["missing_bases"] = int(rng.integers(0, 500))
            bf = rng.dirichlet([30, 20, 20, 30])
            g.attrs["base_freq"] = bf
            for k in kmers:
                d = g.create_dataset(str(k), data=rng.integers(0, 1 << 63, size=s64 * bbits, dtype=np.int64).astype(np.uint64),
                                     dtype="uint64")
                d.attrs["kmer-size"] = k
        if with_random:
            r = f.create_group("random")
            r.attrs["k_min"] = 13
            r.attrs["use_rc"] = True
            r.create_dataset("table_keys", data=np.asarray([n.encode() for n in names]))
            r.create_dataset("table_values", data=np.arange(len(names), dtype=np.uint16) % 2)
            r.create_dataset("centroids", data=rng.random((2, 4)))
            m = r.create_group("matches")
            for k in kmers:
                m.create_dataset(str(k), data=rng.random(4).astype(np.float32))


def main():
    ns = {"h5py": h5py, "np": np, "os": os, "sys": sys}
    extract_functions(os.path.join(REF, "PopPUNK", "sketchlib.py"), NAMES, ns)
    dbdir = os.path.join(HERE, "db")
    os.makedirs(dbdir, exist_ok=True)
    a_names = ["s_alpha", "s_beta", "s_gamma", "s_delta", "s_eps"]
    b_names = ["q_one", "q_two", "a_first"]                 # 'a_first' sorts before every name of a
    write_db(os.path.join(dbdir, "a.h5"), a_names, 1, True)
    write_db(os.path.join(dbdir, "b.h5"), b_names, 2, False)
    ops = {"inputs": {"a": listing(os.path.join(dbdir, "a.h5")), "b": listing(os.path.join(dbdir, "b.h5"))}}
    tmp = tempfile.mkdtemp()
    try:
        # the prefix layout <dir>/<basename>.h5 the non-full_names forms expect
        for nm in ("a", "b"):
            os.makedirs(os.path.join(tmp, nm))
            shutil.copy(os.path.join(dbdir, nm + ".h5"), os.path.join(tmp, nm, nm + ".h5"))
        os.makedirs(os.path.join(tmp, "joined"))
        os.makedirs(os.path.join(tmp, "pruned"))
        pa, pb = os.path.join(tmp, "a"), os.path.join(tmp, "b")
        # joinDBs, prefixes (PopPUNK/assign.py:741 shape, without the random update)
        ns["joinDBs"](pa, pb, os.path.join(tmp, "joined"))
        ops["join_ab"] = listing(os.path.join(tmp, "joined", "joined.h5"))
        # joinDBs the other way round (no /random in the first database), full names
        ns["joinDBs"](os.path.join(pb, "b.h5"), os.path.join(pa, "a.h5"), os.path.join(tmp, "ba"), full_names=True)
        ops["join_ba_full"] = listing(os.path.join(tmp, "ba.h5"))
        # removeFromDB, prefixes: writes <out>/<out>.tmp.h5 (the caller renames it, PopPUNK/assign.py:800-806)
        ns["removeFromDB"](pa, os.path.join(tmp, "pruned"), ["s_beta", "s_eps", "not_there"])
        ops["remove_a"] = listing(os.path.join(tmp, "pruned", "pruned.tmp.h5"))
        # removeFromDB, full names, nothing to remove, no /random
        ns["removeFromDB"](os.path.join(pb, "b.h5"), os.path.join(tmp, "b_same.h5"), [], full_names=True)
        ops["remove_none_b_full"] = listing(os.path.join(tmp, "b_same.h5"))
        kmers, s, cp = ns["readDBParams"](pa)
        ops["readDBParams_a"] = {"kmers": [int(k) for k in kmers], "sketch_size": int(s), "codon_phased": bool(cp)}
        ops["getSeqsInDb_a"] = ns["getSeqsInDb"](os.path.join(pa, "a.h5"))
        ops["getSeqsInDb_joined"] = ns["getSeqsInDb"](os.path.join(tmp, "joined", "joined.h5"))
        gl, amb = ns["get_database_statistics"](pa)
        ops["get_database_statistics_a"] = {"genome_lengths": [int(x) for x in gl], "ambiguous_bases": [int(x) for x in amb]}
    finally:
        shutil.rmtree(tmp)
    with open(os.path.join(HERE, "db_ops.json"), "w") as f:
        json.dump(ops, f, indent=0, sort_keys=True)
    print("wrote db/a.h5, db/b.h5, db_ops.json")


if __name__ == "__main__":
    main()
