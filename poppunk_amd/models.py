"""The boundary-assignment half of PopPUNK's refine/threshold models on the MI355X.

Mirrors what `RefineFit.assign` / `RefineFit.apply_threshold` (PopPUNK/models.py:956-994,
:1065-1091) do around kernel 2, and the hand-off `construct_network_from_assignments` makes to
`generateTuples` (PopPUNK/network.py:1170-1184).  Fitting itself (the optimiser walking the
boundary, models.py:846-954) is outside the hot path; a fitted boundary is the input here.

  * `assign(X)`          : `poppunk_refine.assignThreshold(X/self.scale, slope, x_max, y_max)` with
                           the slope -> (x_max, y_max) mapping of the reference
                           (2: optimal_x/optimal_y, 0: core_boundary/0, 1: 0/accessory_boundary)
  * `assign_dev(dist_t)` : the same on a resident CUDA matrix (float32 division on the device is
                           the IEEE division numpy does)
  * `edges(X)`           : assign -> generateTuples(y, within_label = -1)
  * `edges_from_sketches(db, ...)` : distances, X/scale, boundary and edge compaction fused in
                           one pass (`engine.dist_edges`): the distance matrix never exists
"""
import numpy as np

from . import engine, poppunk_refine

WITHIN_LABEL = -1          # RefineFit.within_label, PopPUNK/models.py:801


class RefineBoundary:
    """A fitted refine/threshold boundary (the state `RefineFit.assign` reads)."""

    def __init__(self, scale=(1.0, 1.0), slope=2, optimal_x=None, optimal_y=None,
                 core_boundary=None, accessory_boundary=None, threads=1):
        self.scale = np.asarray(scale, dtype=np.float32)
        self.slope = int(slope)
        self.optimal_x, self.optimal_y = optimal_x, optimal_y
        self.core_boundary, self.accessory_boundary = core_boundary, accessory_boundary
        self.threads = threads
        self.within_label = WITHIN_LABEL
        self.fitted = optimal_x is not None or core_boundary is not None or accessory_boundary is not None
        self.threshold = False

    @classmethod
    def from_threshold(cls, threshold, dtype=np.float32):
        """RefineFit.apply_threshold (models.py:956-994): vertical line at core = threshold,
        scale (1, 1)."""
        b = cls(scale=np.array([1, 1], dtype=dtype), slope=0, optimal_x=threshold, optimal_y=np.nan,
                core_boundary=threshold, accessory_boundary=np.nan)
        b.threshold = True
        return b

    def _line(self, slope):
        if slope == 2:
            return self.optimal_x, self.optimal_y
        if slope == 0:
            return self.core_boundary, 0
        if slope == 1:
            return 0, self.accessory_boundary
        raise RuntimeError("slope must be 0, 1 or 2")

    def assign(self, X, slope=None):
        """models.py:1065-1091."""
        if not self.fitted:
            raise RuntimeError("Trying to assign using an unfitted model")
        if slope is None:
            slope = self.slope
        x_max, y_max = self._line(slope)
        return poppunk_refine.assignThreshold(X / self.scale, slope, x_max, y_max, self.threads)

    def assign_dev(self, dist_t, slope=None):
        if not self.fitted:
            raise RuntimeError("Trying to assign using an unfitted model")
        import torch
        if slope is None:
            slope = self.slope
        x_max, y_max = self._line(slope)
        scaled = dist_t / torch.as_tensor(self.scale, device=dist_t.device)
        return engine.assign_threshold_dev(scaled, slope, x_max, y_max)

    def edges(self, X, self_comparison=True, num_ref=0, int_offset=0, slope=None):
        """assign -> generateTuples(assignments, within_label, self, num_ref, int_offset):
        the connections construct_network_from_assignments passes on (network.py:1180-1184)."""
        y = self.assign(X, slope)
        return poppunk_refine.generateTuples(y, self.within_label, self=self_comparison,
                                             num_ref=num_ref, int_offset=int_offset)

    def edges_from_sketches(self, db, qry_db, kmers, random_tbl, slope=None, **kw):
        """CUDA int64 [n_edges, 2]: (i, j) of every pair strictly within the boundary, in the
        order generateTuples would emit them."""
        if not self.fitted:
            raise RuntimeError("Trying to assign using an unfitted model")
        if slope is None:
            slope = self.slope
        x_max, y_max = self._line(slope)
        return engine.dist_edges(db, qry_db, kmers, random_tbl, slope=slope, x_max=float(x_max),
                                 y_max=float(y_max), scale=tuple(float(v) for v in self.scale),
                                 inclusive=False, **kw)


# ---- the lineage models' neighbour matrices (PopPUNK/models.py:1095-1385) -----------------------------
EPSILON = 1e-10            # PopPUNK/models.py:75: sparse matrices hold no explicit zeros


class LineageRanks:
    """The state `LineageFit` builds and reads -- `nn_dists` (the neighbours at the search depth) and
    `lower_rank_dists[rank]`, scipy COO matrices -- with the same steps behind the same method names:

      * `fit(X)`                   : models.py:1188-1237 on a long-form distance matrix:
                                     longToSquare -> get_kNN_distances -> lowerRank per rank
      * `fit_from_database(...)`   : the same matrices straight from the sketches
                                     (`pp_sketchlib.queryDatabaseKNN`): no distance matrix in any form
      * `extend(qqDists, qrDists)` : models.py:1334-1385, queries added to a fitted model
      * `assign(rank)` / `edge_weights(rank)` : models.py:1300-1332
    As in the reference a rank equal to the search depth without link filtering is stored as it is
    (`reduce_rank`, models.py:1095-1107) and distances below 1e-10 are raised to it (`__save_sparse__`)."""

    def __init__(self, ranks, max_search_depth, reciprocal_only=False, count_unique_distances=False,
                 lineage_resolution=EPSILON, dist_col=0, threads=1):
        self.ranks = sorted(int(r) for r in ranks)
        if self.ranks[0] < 1:
            raise RuntimeError("Rank must be at least 1")
        self.max_search_depth = max(int(max_search_depth), self.ranks[-1] + 5)      # models.py:1126
        self.reciprocal_only = bool(reciprocal_only)
        self.count_unique_distances = bool(count_unique_distances)
        self.resolution = lineage_resolution
        self.dist_col = int(dist_col)
        self.threads = threads
        self.nn_dists = None
        self.lower_rank_dists = {}
        self.fitted = False

    @staticmethod
    def _coo(data, row, col, n_samples):
        from scipy.sparse import coo_matrix
        data = np.array(data, dtype=np.float32)
        data[data < EPSILON] = EPSILON
        return coo_matrix((data, (np.asarray(row, dtype=np.int64), np.asarray(col, dtype=np.int64))),
                          shape=(n_samples, n_samples), dtype=np.float32)

    def _ranks_from(self, higher, n_samples):
        i, j, d = higher
        for rank in self.ranks:
            if rank == self.max_search_depth and not self.reciprocal_only and not self.count_unique_distances:
                self.lower_rank_dists[rank] = self._coo(d, i, j, n_samples)
            else:
                li, lj, ld = poppunk_refine.lowerRank_arrays((i, j, d), n_samples, rank, self.reciprocal_only,
                                                             self.count_unique_distances, self.resolution,
                                                             self.threads)
                self.lower_rank_dists[rank] = self._coo(ld, li, lj, n_samples)

    def _search_depth(self, n_samples):
        if self.ranks[-1] >= n_samples:
            raise RuntimeError("Maximum rank must be less than the number of samples: " + str(n_samples))
        return min(self.max_search_depth, n_samples - 1)

    def fit(self, X):
        from . import pp_sketchlib
        X = np.asarray(X)
        n = int(round(0.5 * (1 + np.sqrt(1 + 8 * X.shape[0]))))
        depth = self._search_depth(n)
        square = pp_sketchlib.longToSquare(np.ascontiguousarray(X[:, self.dist_col]), self.threads)
        i, j, d = poppunk_refine.get_kNN_distances(square, depth, self.dist_col, self.threads)
        return self._fitted((np.asarray(i, dtype=np.int64), np.asarray(j, dtype=np.int64),
                             np.asarray(d, dtype=np.float32)), depth, n)

    def fit_from_database(self, db_name, names, klist, random_correct=True, device_id=0):
        from . import pp_sketchlib
        n = len(names)
        depth = self._search_depth(n)
        if depth > 32:
            raise RuntimeError("neighbours straight from the sketches: search depth <= 32 (fit(X) has no limit)")
        return self._fitted(pp_sketchlib.queryDatabaseKNN(db_name, names, klist, depth, self.dist_col,
                                                          random_correct, device_id=device_id), depth, n)

    def _fitted(self, higher, depth, n):
        self.nn_dists = self._coo(higher[2], higher[0], higher[1], n)
        self._ranks_from(higher, n)
        self.fitted = True
        return self.assign(self.ranks[0])

    def extend(self, qqDists, qrDists):
        from . import pp_sketchlib
        if not self.fitted:
            raise RuntimeError("Trying to extend an unfitted model")
        qq = pp_sketchlib.longToSquare(np.ascontiguousarray(np.asarray(qqDists)[:, self.dist_col]), self.threads)
        qq[qq < EPSILON] = EPSILON
        n_ref, n_query = self.nn_dists.shape[0], qq.shape[1]
        qr = np.asarray(qrDists)[:, self.dist_col].reshape(n_query, n_ref).T
        qr = np.where(qr < EPSILON, np.float32(EPSILON), qr).astype(np.float32)
        higher = poppunk_refine.extend_arrays((self.nn_dists.row, self.nn_dists.col, self.nn_dists.data), qq, qr,
                                              self.max_search_depth, self.threads)
        self.nn_dists = self._coo(higher[2], higher[0], higher[1], n_ref + n_query)
        self._ranks_from(higher, n_ref + n_query)
        return self.assign(self.ranks[0])

    def extend_from_databases(self, ref_db_name, query_db_name, rList, qList, klist, random_correct=True,
                              device_id=0):
        """`extend` for queries that are still sketches (`pp_sketchlib.extendFromDatabases`): the same matrices,
        without the query x reference and query x query distance matrices."""
        from . import pp_sketchlib
        if not self.fitted:
            raise RuntimeError("Trying to extend an unfitted model")
        if self.max_search_depth > 32:
            raise RuntimeError("neighbours straight from the sketches: search depth <= 32 (extend(qq, qr) has no limit)")
        higher = pp_sketchlib.extendFromDatabases((self.nn_dists.row, self.nn_dists.col, self.nn_dists.data),
                                                  ref_db_name, query_db_name, rList, qList, klist,
                                                  self.max_search_depth, self.dist_col, random_correct,
                                                  device_id=device_id)
        n = len(rList) + len(qList)
        self.nn_dists = self._coo(higher[2], higher[0], higher[1], n)
        self._ranks_from((higher[0], higher[1], self.nn_dists.data), n)
        return self.assign(self.ranks[0])

    def assign(self, rank):
        if not self.fitted:
            raise RuntimeError("Trying to assign using an unfitted model")
        m = self.lower_rank_dists[rank]
        return list(zip(m.row.tolist(), m.col.tolist()))

    def edge_weights(self, rank):
        if not self.fitted:
            raise RuntimeError("Trying to get weights from an unfitted model")
        return self.lower_rank_dists[rank].data
