"""On-disk hand-off formats either side of the distance path.

* `<prefix>.dists.pkl` + `<prefix>.dists.npy` -- what `--create-db` leaves for `--fit-model`
  (PopPUNK/utils.py:135-196): a pickle of `[rlist, qlist, self]` and the float32 [n_pairs, 2]
  matrix.  Same file contents, so either side can be PopPUNK itself.
* the edge list hand-off to `construct_network_from_edge_list` (PopPUNK/network.py:734-864):
  a list/array of (i, j) vertex indices into rlist (+ qlist).  graph-tool is not part of this
  image, so `clusters_from_edges` gives the connected components (what `printClusters`
  derives from the graph, network.py:1529) with scipy, for validating an edge list end to end.
"""
import pickle
import sys

import numpy as np


def storePickle(rlist, qlist, self, X, pklName):
    """PopPUNK/utils.py:135-157."""
    with open(pklName + ".pkl", "wb") as pickle_file:
        pickle.dump([rlist, qlist, self], pickle_file)
    if isinstance(X, np.ndarray):
        np.save(pklName + ".npy", X)


def readPickle(pklName, enforce_self=False, distances=True):
    """PopPUNK/utils.py:160-196 (including the message + exit(1) on an incomplete self DB)."""
    with open(pklName + ".pkl", "rb") as pickle_file:
        rlist, qlist, self = pickle.load(pickle_file)
        if enforce_self and (not self or rlist != qlist):
            sys.stderr.write("Old distances " + pklName + ".npy not complete\n")
            sys.exit(1)
    X = np.load(pklName + ".npy") if distances else None
    return rlist, qlist, self, X


def clusters_from_edges(n_vertices, edges):
    """Connected-component label per vertex for an (i, j) edge array (int64 [m, 2])."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    g = coo_matrix((np.ones(len(e), dtype=np.int8), (e[:, 0], e[:, 1])), shape=(n_vertices, n_vertices))
    n_comp, labels = connected_components(g, directed=False)
    return n_comp, labels
