"""Sketch database files for the distance engine.

The reference stores sketches in `<prefix>/<basename>.h5` with one uint64 dataset
per (sample, k) under /sketches/<name>/<k> and the attributes `sketchsize64`,
`bbits`, `kmers` ... on the sample group (PopPUNK/web.py:14-61; readers
PopPUNK/sketchlib.py:109-195).  Sketching itself (pp_sketchlib.constructDatabase)
is out of scope, and h5py is not part of this image, so the primary on-disk form
here is a flat `<prefix>/<basename>.npz` with the same content:

    names         str   [n]
    kmers         int32 [nk]
    sketches      uint64 [n, nk, sketchsize64*bbits]
    sketchsize64, bbits  int
    random_table  float32 [nk, n_clu, n_clu]   (optional; the /random group)
    clusters      uint16 [n]                   (optional; per-sample cluster id)

`load()` also reads the reference's .h5 layout when h5py is importable (sketches
only: the internal layout of pp-sketchlib's /random group is not documented in
the reference tree, so a .h5 database is loaded without a random-match table), and
`python -m poppunk_amd.sketchdb <prefix>/<basename>` converts a .h5 database to .npz in any
interpreter that has h5py (no GPU or torch needed).
"""
import os
import sys

import numpy as np


def db_file(prefix, ext):
    return prefix + ext


def save_npz(db_name, names, kmers, sketches, sketchsize64, bbits, random_table=None,
             clusters=None):
    """db_name is the reference's `<prefix>/<basename>` (no extension)."""
    os.makedirs(os.path.dirname(db_name) or ".", exist_ok=True)
    payload = dict(names=np.asarray(names, dtype=str), kmers=np.asarray(kmers, dtype=np.int32),
                   sketches=np.ascontiguousarray(sketches, dtype=np.uint64),
                   sketchsize64=np.int32(sketchsize64), bbits=np.int32(bbits))
    if random_table is not None:
        payload["random_table"] = np.asarray(random_table, dtype=np.float32)
    if clusters is not None:
        payload["clusters"] = np.asarray(clusters, dtype=np.uint16)
    np.savez(db_name + ".npz", **payload)


class LoadedSketches:
    def __init__(self, names, kmers, sketches, sketchsize64, bbits, random_table, clusters):
        self.names = names
        self.kmers = kmers
        self.sketches = sketches
        self.sketchsize64 = sketchsize64
        self.bbits = bbits
        self.random_table = random_table
        self.clusters = clusters


def _select_kmers(db_kmers, klist):
    db_kmers = [int(k) for k in db_kmers]
    idx = []
    for k in klist:
        if int(k) not in db_kmers:
            raise RuntimeError("k-mer length %d not found in sketch database (has %s)"
                               % (int(k), db_kmers))
        idx.append(db_kmers.index(int(k)))
    return idx


def _load_npz(path, names, klist):
    with np.load(path, allow_pickle=False) as z:
        db_names = [str(x) for x in z["names"]]
        pos = {nm: i for i, nm in enumerate(db_names)}
        missing = [nm for nm in names if nm not in pos]
        if missing:
            raise RuntimeError("samples not found in sketch database %s: %s"
                               % (path, ", ".join(missing[:5])))
        kidx = _select_kmers(z["kmers"], klist)
        rows = np.asarray([pos[nm] for nm in names], dtype=np.int64)
        sk = z["sketches"][rows][:, kidx, :]
        tbl = z["random_table"][kidx] if "random_table" in z.files else None
        clu = z["clusters"][rows] if "clusters" in z.files else None
        return LoadedSketches(list(names), np.asarray(klist, dtype=np.int32),
                              np.ascontiguousarray(sk), int(z["sketchsize64"]), int(z["bbits"]),
                              tbl, clu)


def _load_h5(path, names, klist):
    import h5py  # optional
    with h5py.File(path, "r") as f:
        grp = f["sketches"]
        first = grp[names[0]]
        s64 = int(first.attrs["sketchsize64"])
        bbits = int(first.attrs["bbits"])
        sk = np.empty((len(names), len(klist), s64 * bbits), dtype=np.uint64)
        for i, nm in enumerate(names):
            if nm not in grp:
                raise RuntimeError("sample %s not found in sketch database %s" % (nm, path))
            for j, k in enumerate(klist):
                sk[i, j] = grp[nm][str(int(k))][:]
        if "random" in f:
            sys.stderr.write("poppunk_amd: %s has a /random group, but its layout is internal to "
                             "pp-sketchlib; proceeding without random-match correction\n" % path)
        return LoadedSketches(list(names), np.asarray(klist, dtype=np.int32), sk, s64, bbits,
                              None, None)


def convert_h5_to_npz(db_name, out_name=None):
    """Rewrite the reference's `<db_name>.h5` (PopPUNK/web.py:14-61 layout) as `<out_name>.npz`
    (default: next to it) with every sample and every k-mer length it holds.  Needs h5py, not a
    GPU: `python -m poppunk_amd.sketchdb <prefix>/<basename>` runs it in whatever interpreter has
    h5py.  The /random group is not carried over (its layout is internal to pp-sketchlib)."""
    import h5py  # optional
    with h5py.File(db_name + ".h5", "r") as f:
        grp = f["sketches"]
        names = sorted(grp.keys())
        if not names:
            raise RuntimeError("no samples in %s.h5" % db_name)
        kmers = sorted(int(k) for k in grp[names[0]].attrs["kmers"])
    loaded = _load_h5(db_name + ".h5", names, kmers)
    save_npz(out_name or db_name, loaded.names, loaded.kmers, loaded.sketches, loaded.sketchsize64,
             loaded.bbits)
    return len(names), kmers


def load(db_name, names, klist):
    """Load `names` x `klist` from `<db_name>.npz` (preferred) or `<db_name>.h5`."""
    names = [str(n) for n in names]
    klist = [int(k) for k in np.asarray(klist).ravel()]
    if len(names) == 0:
        raise RuntimeError("no sample names given")
    if os.path.exists(db_name + ".npz"):
        return _load_npz(db_name + ".npz", names, klist)
    if os.path.exists(db_name + ".h5"):
        try:
            return _load_h5(db_name + ".h5", names, klist)
        except ImportError:
            raise RuntimeError("reading %s.h5 needs h5py, which is not installed; convert the database with "
                               "`python -m poppunk_amd.sketchdb %s` in an interpreter that has it" % (db_name, db_name))
    raise RuntimeError("sketch database %s(.npz|.h5) not found" % db_name)


if __name__ == "__main__":
    if len(sys.argv) not in (2, 3):
        sys.exit("usage: python -m poppunk_amd.sketchdb <prefix>/<basename> [<out prefix>/<basename>]   (.h5 -> .npz)")
    n_done, k_done = convert_h5_to_npz(sys.argv[1], sys.argv[2] if len(sys.argv) == 3 else None)
    print("wrote %s.npz: %d samples, k = %s" % (sys.argv[2] if len(sys.argv) == 3 else sys.argv[1], n_done, k_done))
