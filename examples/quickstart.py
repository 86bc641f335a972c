#!/usr/bin/env python3
"""End-to-end walk through the path this repository replaces, on synthetic sketches:

  sketch database (.h5, the reference's layout)               PopPUNK/web.py:14-61
    -> PopPUNK.sketchlib.queryDatabase(...)  (core, accessory) PopPUNK/sketchlib.py:475-632
    -> store the distances the way PopPUNK does                PopPUNK/utils.py:135-157
    -> a refine / threshold boundary: assignments, edge list   PopPUNK/models.py:1065-1091,
                                                               PopPUNK/network.py:1180-1184
    -> clusters = connected components of the edge list
  and the same edge list again from the FUSED call, in which no distance matrix is ever stored -- on resident
  sketches and as one call from the database files -- then distance QC and the lineage models' neighbour matrices.

    python examples/quickstart.py [n_genomes] [workdir]          # needs an MI355X
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poppunk_amd import distfile, engine, models, sketchdb, sketchlib, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    work = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="ppk_quickstart_")
    kmers = np.asarray(synth.DEFAULT_KMERS, dtype=np.int32)          # 13, 17, 21, 25, 29

    # 1. a sketch database on disk, in the reference's HDF5 layout (normally written by `poppunk --create-db`)
    sketches, member = synth.make_sketches(n, kmers, cluster_size=40)
    names = ["genome_%04d" % i for i in range(n)]
    db_prefix = os.path.join(work, "example_db")
    os.makedirs(db_prefix, exist_ok=True)
    sketchdb.save_h5(os.path.join(db_prefix, os.path.basename(db_prefix)), names, kmers, sketches, 16, 14,
                     random_table=synth.random_match_table(kmers), clusters=np.zeros(n, dtype=np.uint16))

    # 2. all-vs-all core / accessory distances: the call PopPUNK makes
    X = sketchlib.queryDatabase(names, names, db_prefix, db_prefix, kmers, self=True)
    assert X.shape == (n * (n - 1) // 2, 2) and X.dtype == np.float32
    print("distances: %d pairs, core %.4f +- %.4f, accessory %.4f +- %.4f"
          % (X.shape[0], X[:, 0].mean(), X[:, 0].std(), X[:, 1].mean(), X[:, 1].std()))

    # 3. stored as PopPUNK stores them (<prefix>.dists.pkl + .npy)
    distfile.storePickle(names, names, True, X, os.path.join(db_prefix, "example_db.dists"))
    rlist, qlist, self_flag, X2 = distfile.readPickle(os.path.join(db_prefix, "example_db.dists"))
    assert self_flag and np.array_equal(X, X2)

    # 4. a fitted boundary (here: the triangle through the 6 % quantile) -> assignments -> edges -> clusters
    x_max, y_max = synth.boundary_for_quantile(X, 0.06)
    boundary = models.RefineBoundary(scale=(1.0, 1.0), slope=2, optimal_x=x_max, optimal_y=y_max)
    y = boundary.assign(X)                                  # -1 within, 0 on the line, +1 outside
    edges = np.asarray(boundary.edges(X), dtype=np.int64).reshape(-1, 2)
    n_clusters, labels = distfile.clusters_from_edges(n, edges)
    print("boundary (%.4f, %.4f): %d within-strain pairs -> %d clusters (%d synthetic lineages)"
          % (x_max, y_max, len(edges), n_clusters, len(set(member.tolist()))))
    assert (y == -1).sum() == len(edges)

    # 5. the same edge list from the fused call: distance, scaling, boundary and compaction in one pass
    loaded = sketchdb.load(os.path.join(db_prefix, os.path.basename(db_prefix)), names, kmers)
    db = engine.SketchDB(loaded.sketches, loaded.sketchsize64, loaded.bbits, clusters=loaded.clusters)
    fused, n_failed = boundary.edges_from_sketches(db, None, kmers, loaded.random_table)
    fused = fused.cpu().numpy()
    db.close()
    assert np.array_equal(fused, edges), "fused edge list differs"
    print("fused distance -> boundary -> edge list: identical (%d edges), %d failed fits" % (len(fused), n_failed))

    # 6. ... and as ONE call from the database files to a numpy edge list (every GPU in PPK_DEVICES takes a band)
    from poppunk_amd import pp_sketchlib
    base = os.path.join(db_prefix, os.path.basename(db_prefix))
    one_call = pp_sketchlib.queryDatabaseEdges(base, base, names, names, kmers, boundary.slope, x_max, y_max,
                                               scale=boundary.scale, inclusive=False)
    assert np.array_equal(one_call, edges), "queryDatabaseEdges differs"

    # 7. QC of the distances (PopPUNK/qc.py:295-369) and the lineage models' neighbour matrices
    #    (get_kNN_distances -> lowerRank, PopPUNK/models.py:1177,:1215-1222)
    from poppunk_amd import poppunk_refine, qc
    kept, failed = qc.qcDistMat(X, names, names, db_prefix, {"max_pi_dist": 0.5, "max_a_dist": 0.6, "prop_zero": 0.05})
    square = pp_sketchlib.longToSquare(np.ascontiguousarray(X[:, 0]))
    nn = poppunk_refine.get_kNN_distances(square, 3)
    rank1 = poppunk_refine.lowerRank(nn, n, 1, False, False, 1e-5)
    print("distance QC keeps %d of %d samples; rank-3 neighbour matrix %d entries, rank 1 %d"
          % (len(kept), n, len(nn[0]), len(rank1[0])))
    print("files in", db_prefix)


if __name__ == "__main__":
    main()
