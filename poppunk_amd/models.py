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
