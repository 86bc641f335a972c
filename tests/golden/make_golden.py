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
  boundary_refine.npz  the reference's own pure-Python statement of kernel 2 --
                       withinBoundary / iter_tuples of test/test-refine.py:10-38 -- run on
                       the grid of test-refine.py:47-50 and on seeded stand-ins for its
                       unseeded random matrices (:64-66), plus the per-offset expected edge
                       sets of its thresholdIterate1D/2D sections (:84-138) with
                       withinBoundary in the role poppunk_refine.assignThreshold plays there.
                       This is the pin of the kernel-2 oracle (oracle/ppk_oracle.c).
  qc.json, qc_autodist.npz
                       PopPUNK.qc.qcDistMat / prune_edges / autoDistFind (PopPUNK/qc.py:238-369,:419-468)
                       on small seeded matrices (see golden_qc for what stands in for the compiled
                       poppunk_refine.generateTuples)
boundary_known_answers.json is NOT generated here and pins nothing: it holds values
hand-transcribed from SURVEY.md Appendix B and is kept only as a cross-check.
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

    # The same reference function once more with ONLY the solver's stopping tolerances tightened: its
    # default ftol = 1e-8 leaves ~5e-6 of solver error in the parameters, more than the 1e-6 this path
    # is held to.  Model, Jacobian, start point and bounds stay the reference's own (the function body
    # runs unchanged; `optimize` in its namespace forwards to scipy with xtol = ftol = gtol = 1e-15).
    class TightOptimize:
        @staticmethod
        def least_squares(*a, **k):
            k.update(xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=100000)
            return optimize.least_squares(*a, **k)
    ns_t = {"np": np, "optimize": TightOptimize, "sys": sys}
    extract_functions(os.path.join(REF, "PopPUNK", "sketchlib.py"), ["fitKmerCurve"], ns_t)
    fit_tight = ns_t["fitKmerCurve"]
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
            core_t, acc_t = fit_tight(y, klist, jac_mat)
            # unconstrained OLS, to record whether a bound was active in the reference fit
            A = np.vstack([np.ones_like(klist, dtype=np.float64), klist.astype(np.float64)]).T
            icpt, slope = np.linalg.lstsq(A, np.log(y), rcond=None)[0]
            cases.append({"klist": klist.tolist(), "jaccard": y.tolist(),
                          "core": float(core), "accessory": float(acc),
                          "core_tight": float(core_t), "accessory_tight": float(acc_t),
                          "interior": bool(icpt < 0 and slope < 0)})
    # exact model points: fitKmerCurve((1-0.1)*(1-0.02)**k) -> [0.02, 0.1]
    klist = np.arange(13, 30, 4)
    jac_mat = -np.hstack((np.ones((klist.shape[0], 1)), klist.reshape(-1, 1)))
    y = (1 - 0.1) * (1 - 0.02) ** klist.astype(np.float64)
    core, acc = fit(y, klist, jac_mat)
    core_t, acc_t = fit_tight(y, klist, jac_mat)
    cases.append({"klist": klist.tolist(), "jaccard": y.tolist(), "core": float(core),
                  "accessory": float(acc), "core_tight": float(core_t), "accessory_tight": float(acc_t),
                  "interior": True})
    with open(os.path.join(HERE, "fit_kmer_curve.json"), "w") as f:
        json.dump({"source": "PopPUNK/sketchlib.py:635-670 fitKmerCurve, run by make_golden.py; *_tight: the same "
                             "function with only scipy's stopping tolerances set to 1e-15",
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


def golden_qc():
    """PopPUNK.qc.qcDistMat / prune_edges / autoDistFind (PopPUNK/qc.py:238-369,:419-468) run on small
    seeded matrices.  qcDistMat calls the compiled poppunk_refine.generateTuples, which cannot be built
    here; in its role runs oracle.generate_tuples -- this repo's restatement of src/boundary.cpp:97-123,
    itself pinned by boundary_refine.npz -- and every other line is the reference's own."""
    from collections import Counter
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle

    class Refine:
        @staticmethod
        def generateTuples(assignments, within_label, self=True, num_ref=0, int_offset=0):
            e = oracle.generate_tuples(np.asarray(assignments, dtype=np.int32), within_label, self=self,
                                       num_ref=num_ref, int_offset=int_offset)
            return [tuple(int(v) for v in row) for row in np.asarray(e).tolist()]

    ns = {"np": np, "sys": sys, "Counter": Counter, "poppunk_refine": Refine}
    extract_functions(os.path.join(REF, "PopPUNK", "qc.py"), ["qcDistMat", "prune_edges", "autoDistFind"], ns)
    rng = np.random.Generator(np.random.PCG64(20260930))
    out = {"source": "PopPUNK/qc.py:238-369,:419-468 qcDistMat / prune_edges / autoDistFind, run by "
                     "make_golden.py with oracle.generate_tuples in the role of poppunk_refine.generateTuples",
           "qcDistMat": [], "prune_edges": [], "autoDistFind": []}

    def population(n_ref, n_qry, bad, zero):
        """distances of n samples; `bad` samples sit far from everything, `zero` pairs are duplicates"""
        rows = n_ref * (n_ref - 1) // 2 if n_qry == 0 else n_ref * n_qry
        d = np.stack([rng.uniform(0.001, 0.02, rows), rng.uniform(0.01, 0.3, rows)], axis=1).astype(np.float32)
        pairs = ([(i, j) for i in range(n_ref) for j in range(i + 1, n_ref)] if n_qry == 0
                 else [(r, n_ref + q) for q in range(n_qry) for r in range(n_ref)])
        for row, (i, j) in enumerate(pairs):
            if i in bad or j in bad:
                d[row] = (rng.uniform(0.06, 0.2), rng.uniform(0.55, 0.9))
            if (i, j) in zero:
                d[row, rng.integers(0, 2)] = 0.0
        return d

    stderr = sys.stderr
    sys.stderr = open(os.devnull, "w")
    try:
        cases = [  # n_ref, n_qry, bad samples, zero pairs, prop_zero
            (12, 0, {3}, set(), 1.0),
            (20, 0, {0, 7, 19}, {(1, 2), (1, 5), (1, 9), (4, 6)}, 0.1),
            (30, 0, set(), {(2, 3), (2, 4), (2, 5), (2, 6), (8, 9)}, 0.1),
            (15, 0, {5, 6}, {(5, 6), (0, 1)}, 0.05),
            (8, 5, {9, 12}, {(0, 8), (1, 8), (2, 8)}, 0.2),
            (10, 6, {2}, set(), 0.5),
            (6, 4, set(), set(), 0.05),
        ]
        for n_ref, n_qry, bad, zero, prop_zero in cases:
            d = population(n_ref, n_qry, bad, zero)
            refs = ["r%02d" % i for i in range(n_ref)]
            qrys = refs if n_qry == 0 else ["q%02d" % i for i in range(n_qry)]
            qc = {"max_pi_dist": 0.05, "max_a_dist": 0.5, "prop_zero": prop_zero}
            kept, failed = ns["qcDistMat"](d, refs, qrys, "unused_db", qc)
            out["qcDistMat"].append({"refs": refs, "queries": qrys, "dist": d.tolist(), "qc_dict": qc,
                                     "retained": list(kept), "failed": {k: list(v) for k, v in failed.items()}})
        for n_nodes, n_edges, query_start, min_count, allow, pre in (
                (10, 12, 10, 1, True, None), (10, 20, 6, 1, False, None), (12, 30, 7, 3, True, {2}),
                (9, 15, 4, 2, False, {0, 8}), (16, 40, 16, 4, True, None), (5, 0, 3, 1, True, None)):
            e = set()
            while len(e) < n_edges:
                a, b = sorted(int(v) for v in rng.integers(0, n_nodes, 2))
                if a != b:
                    e.add((a, b))
            edges = sorted(e)
            failed = ns["prune_edges"](list(edges), query_start, failed=None if pre is None else set(pre),
                                       min_count=min_count, allow_ref_ref=allow)
            out["prune_edges"].append({"edges": [list(x) for x in edges], "query_start": query_start,
                                       "min_count": min_count, "allow_ref_ref": allow,
                                       "failed_before": None if pre is None else sorted(pre),
                                       "failed": sorted(int(v) for v in failed)})
        mats = {}
        for i, (rows, r, x, n_out) in enumerate(((8000, 20, 0.2, 60), (8000, 20, 0.2, 0), (12000, 30, 0.1, 200),
                                                 (6000, 10, 0.5, 20))):
            d = np.stack([rng.gamma(4.0, 0.002, rows), rng.gamma(4.0, 0.03, rows)], axis=1).astype(np.float32)
            if n_out:
                idx = rng.choice(rows, n_out, replace=False)
                d[idx, 0] *= rng.uniform(4, 6, n_out).astype(np.float32)
                d[idx[: n_out // 2], 1] *= rng.uniform(5, 8, n_out // 2).astype(np.float32)
            max_pi, max_a = ns["autoDistFind"](d, {"x": x, "r": r})
            mats["dist%d" % i] = d
            out["autoDistFind"].append({"dist": "qc_autodist.npz:dist%d" % i, "r": r, "x": x,
                                        "max_pi": float(max_pi), "max_a": float(max_a)})
    finally:
        sys.stderr.close()
        sys.stderr = stderr
    np.savez_compressed(os.path.join(HERE, "qc_autodist.npz"), **mats)
    with open(os.path.join(HERE, "qc.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("qc.json:", len(out["qcDistMat"]), "qcDistMat,", len(out["prune_edges"]), "prune_edges,",
          len(out["autoDistFind"]), "autoDistFind cases")


def golden_refine():
    """test/test-refine.py:10-38.  withinBoundary evaluates the float32 rows with the numpy of
    this interpreter (numpy >= 2: float32 * python float stays float32, un-fused) and calls a row
    'on the line' when |in_tri| < float32 eps; src/boundary.cpp tests == 0 exactly.  The test the
    functions come from asserts equality of the two on the grid (check_res, :59-61) and edge-set
    containment on the random matrix (:68-82).  Rows whose |in_tri| lies inside that eps band but
    is not exactly zero are marked `band` (ambiguous between the two statements); everywhere
    else the fixture is the reference's answer."""
    ns = {"np": np}
    extract_functions(os.path.join(REF, "test", "test-refine.py"), ["withinBoundary", "iter_tuples"], ns)
    within, iter_tuples = ns["withinBoundary"], ns["iter_tuples"]
    eps = np.finfo(np.float32).eps
    out = {}

    def in_tri(d, x_max, y_max, slope):
        # the expression of withinBoundary itself (test-refine.py:14-19), row by row
        v = np.empty(d.shape[0], dtype=np.float64)
        for row in range(d.shape[0]):
            if slope == 2:
                v[row] = d[row, 1] * x_max + d[row, 0] * y_max - x_max * y_max
            elif slope == 0:
                v[row] = d[row, 0] - x_max
            else:
                v[row] = d[row, 1] - y_max
        return v

    # the grid of test-refine.py:47-50
    x = np.arange(0, 1, 0.1, dtype=np.float32)
    y = np.arange(0, 1, 0.1, dtype=np.float32)
    xv, yv = np.meshgrid(x, y)
    grid = np.hstack((xv.reshape(-1, 1), yv.reshape(-1, 1)))
    out["grid"] = grid
    for slope in (0, 1, 2):
        out["grid_assign%d" % slope] = within(grid, 0.5, 0.5, slope).astype(np.float32)
        t = in_tri(grid, 0.5, 0.5, slope)
        out["grid_band%d" % slope] = (np.abs(t) < eps) & (t != 0)

    # seeded stand-ins for the unseeded np.random.rand matrix of :64-66
    rng = np.random.Generator(np.random.PCG64(20260928))
    for samples in (100, 363):
        d = np.array(rng.random((int(0.5 * samples * (samples - 1)), 2)), dtype=np.float32)
        # a few rows exactly on each boundary and a few inside the eps band, so that the
        # `== 0` class and the band are both exercised (0.5 and 0.25 are dyadic: exact in float32)
        d[3] = (0.5, 0.5)          # slope 0, 1: on the line
        d[7] = (0.25, 0.25)        # slope 2 with x_max = y_max = 0.5: 0.125 + 0.125 - 0.25 = 0
        d[11] = (0.5, 0.0)         # slope 0 and slope 2: on the line
        d[13] = (np.nextafter(np.float32(0.5), np.float32(1)), 0.9)   # slope 0: one ulp (6e-8) over, inside the band, not zero
        d[17] = (0.1, np.nextafter(np.float32(0.5), np.float32(0)))   # slope 1: half an ulp-of-1 under
        key = "rand%d" % samples
        out[key] = d
        for slope in (0, 1, 2):
            a = within(d, 0.5, 0.5, slope)
            t = in_tri(d, 0.5, 0.5, slope)
            out["%s_assign%d" % (key, slope)] = a.astype(np.float32)
            out["%s_band%d" % (key, slope)] = (np.abs(t) < eps) & (t != 0)
            out["%s_edges%d" % (key, slope)] = np.asarray(iter_tuples(a, samples),
                                                           dtype=np.int64).reshape(-1, 2)

    # thresholdIterate1D / 2D sections (:84-138) on the 100-sample matrix: the expected edge set
    # per offset is {(i, j): assign <= 0}; withinBoundary stands where the test calls
    # poppunk_refine.assignThreshold
    from math import sqrt
    d = out["rand100"]
    samples = 100
    offsets = [v * sqrt(2) for v in [-0.1, 0.0, 0.1]]
    out["it1d_offsets"] = np.asarray(offsets, dtype=np.float64)
    out["it1d_line"] = np.asarray([0.2, 0.2, 0.3, 0.3], dtype=np.float64)   # x0, y0, x1, y1
    for oi, off in enumerate(offsets):
        xmax = 0.4 + (2 * (off / sqrt(2)))
        a = within(d, xmax, xmax, 2)
        out["it1d_xmax%d" % oi] = np.float64(xmax)
        out["it1d_rows%d" % oi] = np.flatnonzero(a <= 0).astype(np.int64)
        t = in_tri(d, xmax, xmax, 2)
        out["it1d_band%d" % oi] = np.flatnonzero((np.abs(t) < eps) & (t != 0)).astype(np.int64)
    xs = [0.1, 0.2, 0.3]
    out["it2d_xmax"] = np.asarray(xs, dtype=np.float64)
    out["it2d_ymax"] = np.float64(0.2)
    for oi, xm in enumerate(xs):
        a = within(d, xm, 0.2, 2)
        out["it2d_rows%d" % oi] = np.flatnonzero(a <= 0).astype(np.int64)
        t = in_tri(d, xm, 0.2, 2)
        out["it2d_band%d" % oi] = np.flatnonzero((np.abs(t) < eps) & (t != 0)).astype(np.int64)
    out["numpy_version"] = np.asarray(np.__version__)
    out["source"] = np.asarray("test/test-refine.py:10-38 withinBoundary / iter_tuples, executed by "
                               "tests/golden/make_golden.py golden_refine()")
    np.savez_compressed(os.path.join(HERE, "boundary_refine.npz"), **out)
    print("boundary_refine.npz:", {k: (int(out[k].sum()) if out[k].dtype == bool else out[k].shape)
                                   for k in sorted(out) if "band" in k or "edges" in k})


if __name__ == "__main__":
    golden_qc()
    golden_refine()
    golden_prune()
    golden_fit()
    golden_rows()
    golden_sketch()
