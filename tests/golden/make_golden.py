#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the reference checkout.

Run in the BUILD container only (needs /root/reference); the fixtures it writes
are data (inputs + expected outputs) and are committed, the reference is not.

How reference code is executed: the two pure-Python functions on the path that
live in the reference tree -- PopPUNK.sketchlib.fitKmerCurve
(PopPUNK/sketchlib.py:635-670) and PopPUNK.utils.iterDistRows / listDistInts
(PopPUNK/utils.py:199-261) -- are pulled out of their modules with `ast` and
executed under the real numpy / scipy of this image.  Their modules cannot be
imported whole because they import pp_sketchlib / h5py / graph_tool, which are
absent here; nothing is stubbed, only the named function definitions run.

Fixtures written:
  fit_kmer_curve.json  (klist, J_k) -> (core, accessory) from fitKmerCurve
  row_order.json       distMat row -> (ref, query) tables from iterDistRows/listDistInts
  json_sketch.npz      the real sketch test/json_sketch.txt (a data file of the
                       reference's test directory) as uint64 arrays
  prune.json           PopPUNK.qc.prune_distance_matrix / prune_query_distance_matrix
                       (PopPUNK/qc.py:17-135) run on small seeded matrices, with the real
                       iterDistRows / storePickle (PopPUNK/utils.py) they call
boundary_known_answers.json is NOT generated here: src/boundary.cpp needs Eigen,
absent from this image; its values were captured during the survey (SURVEY.md
Appendix B) and are transcribed by hand.
"""
import ast
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def extract_functions(path, names, namespace):
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), namespace)
    missing = [n for n in names if n not in namespace]
    if missing:
        raise RuntimeError("not found in %s: %s" % (path, missing))
    return namespace


def golden_fit():
    from scipy import optimize
    ns = {"np": np, "optimize": optimize, "sys": sys}
    extract_functions(os.path.join(REF, "PopPUNK", "sketchlib.py"), ["fitKmerCurve"], ns)
    fit = ns["fitKmerCurve"]
    rng = np.random.Generator(np.random.PCG64(20260928))
    cases = []
    klists = [np.arange(13, 30, 4), np.arange(13, 29, 3), np.arange(15, 32, 2), np.array([13, 29])]
    for klist in klists:
        jac_mat = -np.hstack((np.ones((klist.shape[0], 1)), klist.reshape(-1, 1)))
        for rep in range(14):
            a = rng.uniform(0.0, 0.6)
            c = rng.uniform(0.0, 0.04)
            noise = rng.choice([0.0, 0.01, 0.05])
            if rep >= 12:        # near-identical pair + noise: a bound of the reference fit goes active
                a, c, noise = 0.0, rng.uniform(0.0, 0.002), 0.02
            y = (1 - a) * (1 - c) ** klist.astype(np.float64)
            y = y * np.exp(rng.normal(0.0, noise, size=y.shape))
            y = np.minimum(y, 1.0)
            core, acc = fit(y, klist, jac_mat)
            # unconstrained OLS, to record whether a bound was active in the reference fit
            A = np.vstack([np.ones_like(klist, dtype=np.float64), klist.astype(np.float64)]).T
            icpt, slope = np.linalg.lstsq(A, np.log(y), rcond=None)[0]
            cases.append({"klist": klist.tolist(), "jaccard": y.tolist(),
                          "core": float(core), "accessory": float(acc),
                          "interior": bool(icpt < 0 and slope < 0)})
    # exact model points: fitKmerCurve((1-0.1)*(1-0.02)**k) -> [0.02, 0.1]
    klist = np.arange(13, 30, 4)
    jac_mat = -np.hstack((np.ones((klist.shape[0], 1)), klist.reshape(-1, 1)))
    y = (1 - 0.1) * (1 - 0.02) ** klist.astype(np.float64)
    core, acc = fit(y, klist, jac_mat)
    cases.append({"klist": klist.tolist(), "jaccard": y.tolist(), "core": float(core),
                  "accessory": float(acc), "interior": True})
    with open(os.path.join(HERE, "fit_kmer_curve.json"), "w") as f:
        json.dump({"source": "PopPUNK/sketchlib.py:635-670 fitKmerCurve, run by make_golden.py",
                   "cases": cases}, f, indent=1)
    print("fit_kmer_curve.json:", len(cases), "cases,",
          sum(c["interior"] for c in cases), "interior")


def golden_rows():
    ns = {}
    extract_functions(os.path.join(REF, "PopPUNK", "utils.py"), ["iterDistRows", "listDistInts"], ns)
    out = {"source": "PopPUNK/utils.py:199-261 iterDistRows/listDistInts, run by make_golden.py",
           "self": [], "nonself": []}
    for n in (2, 3, 4, 5, 7):
        names = ["s%d" % i for i in range(n)]
        out["self"].append({"n": n,
                            "names": [list(t) for t in ns["iterDistRows"](names, names, True)],
                            "ints": [list(t) for t in ns["listDistInts"](names, names, True)]})
    for nr, nq in ((2, 3), (3, 2), (4, 1), (1, 4)):
        r = ["r%d" % i for i in range(nr)]
        q = ["q%d" % i for i in range(nq)]
        out["nonself"].append({"n_ref": nr, "n_query": nq,
                               "names": [list(t) for t in ns["iterDistRows"](r, q, False)],
                               "ints": [list(t) for t in ns["listDistInts"](r, q, False)]})
    with open(os.path.join(HERE, "row_order.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("row_order.json written")


def golden_sketch():
    d = json.load(open(os.path.join(REF, "test", "json_sketch.txt")))
    kmers = sorted(int(k) for k in d if k.isdigit())
    sk = np.stack([np.array(d[str(k)], dtype=np.uint64) for k in kmers])
    np.savez_compressed(os.path.join(HERE, "json_sketch.npz"), kmers=np.array(kmers, dtype=np.int32),
                        sketch=sk, bbits=np.int32(d["bbits"]),
                        sketchsize64=np.int32(d["sketchsize64"]), length=np.int64(d["length"]),
                        missing_bases=np.int64(d["missing_bases"]),
                        bases=np.array(d["bases"], dtype=np.float64))
    print("json_sketch.npz:", sk.shape, "kmers", kmers)


def golden_prune():
    import pickle
    import tempfile
    ns = {"np": np, "sys": sys, "pickle": pickle}
    extract_functions(os.path.join(REF, "PopPUNK", "utils.py"), ["iterDistRows", "storePickle"], ns)
    extract_functions(os.path.join(REF, "PopPUNK", "qc.py"),
                      ["prune_distance_matrix", "prune_query_distance_matrix"], ns)
    rng = np.random.Generator(np.random.PCG64(20260929))
    out = {"source": "PopPUNK/qc.py:17-135 prune_distance_matrix / prune_query_distance_matrix, "
                     "run by make_golden.py", "self": [], "query": []}
    with tempfile.TemporaryDirectory() as tmp:
        for n, n_rm in ((4, 1), (7, 2), (9, 4), (12, 1), (6, 0)):
            names = ["s%d" % i for i in range(n)]
            dist = rng.random((n * (n - 1) // 2, 2)).astype(np.float32)
            remove = [names[i] for i in sorted(rng.choice(n, size=n_rm, replace=False))]
            if n == 7:
                remove.append("not_in_db")      # the reference reports it and carries on
            new_names, new_dist = ns["prune_distance_matrix"](names, remove, dist,
                                                               os.path.join(tmp, "p%d" % n))
            out["self"].append({"names": names, "remove": remove, "dist": dist.tolist(),
                                "new_names": list(new_names), "new_dist": np.asarray(new_dist).tolist()})
        for nr, nq, n_rm in ((3, 4, 1), (2, 5, 2), (4, 3, 0)):
            refs = ["r%d" % i for i in range(nr)]
            qrys = ["q%d" % i for i in range(nq)]
            qr = rng.random((nr * nq, 2)).astype(np.float32)
            assign = rng.integers(-1, 2, size=nr * nq).astype(np.int64)
            remove = set(qrys[i] for i in rng.choice(nq, size=n_rm, replace=False))
            passing, new_qr, new_assign = ns["prune_query_distance_matrix"](refs, qrys, remove, qr, assign)
            out["query"].append({"refs": refs, "queries": qrys, "remove": sorted(remove),
                                 "dist": qr.tolist(), "assign": assign.tolist(),
                                 "passing": list(passing), "new_dist": np.asarray(new_qr).tolist(),
                                 "new_assign": np.asarray(new_assign).tolist()})
    with open(os.path.join(HERE, "prune.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("prune.json:", len(out["self"]), "self cases,", len(out["query"]), "query cases")


if __name__ == "__main__":
    golden_prune()
    golden_fit()
    golden_rows()
    golden_sketch()
