"""Mirror of the distance-matrix pruning helpers of PopPUNK/qc.py on the MI355X.

`prune_distance_matrix` (PopPUNK/qc.py:17-91) rebuilds the long-form distance matrix after
samples have been removed; the reference does it with one Python iteration per distance row
(its own comment: "this seems like it would be slow for a big dist matrix").  Here the kept rows
are gathered by one kernel (`ppk_prune_long`, include/ppk.h).  `prune_query_distance_matrix`
(PopPUNK/qc.py:93-135) keeps the row blocks of the passing queries.

Same names, argument order, messages and return values as the reference functions.
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from .distfile import storePickle


def prune_distance_matrix(refList, remove_seqs_in, distMat, output, device_id=0):
    """PopPUNK/qc.py:17-91.  Returns (newRefList, newDistMat); writes `output`.pkl/.npy like the
    reference (skipped when output is None)."""
    remove_seqs_list = []
    removal = set()
    index_of = {}
    for idx, item in enumerate(refList):
        index_of.setdefault(item, idx)          # the reference stops at the first match
    for to_remove in remove_seqs_in:
        if to_remove in index_of:
            removal.add(index_of[to_remove])
            remove_seqs_list.append(to_remove)
        else:
            sys.stderr.write("Couldn't find " + to_remove + " in database\n")
    remove_seqs = frozenset(remove_seqs_list)

    if len(remove_seqs) > 0:
        sys.stderr.write("Removing " + str(len(remove_seqs)) + " sequences\n")
        keep = np.asarray([i for i in range(len(refList)) if i not in removal], dtype=np.int64)
        newRefList = [refList[i] for i in keep]
        d = np.asarray(distMat)
        if d.dtype != np.float32 or not d.flags.c_contiguous:
            raise TypeError("distMat must be a C-contiguous float32 array")
        n, m = len(refList), len(keep)
        if d.shape[0] != n * (n - 1) // 2:
            raise RuntimeError("distMat does not have one row per pair of refList")
        cols = 1 if d.ndim == 1 else d.shape[1]
        newDistMat = np.zeros((m * (m - 1) // 2,) + d.shape[1:], dtype=np.float32)
        if m > 1:
            rc = _lib.lib().ppk_prune_long(d.ctypes.data_as(C.POINTER(C.c_float)), n, cols,
                                           keep.ctypes.data_as(C.POINTER(C.c_longlong)), m,
                                           int(device_id),
                                           newDistMat.ctypes.data_as(C.POINTER(C.c_float)))
            _lib.check(rc, "ppk_prune_long")
    else:
        newRefList = refList
        newDistMat = distMat

    if output is not None:
        storePickle(newRefList, newRefList, True, newDistMat, output)
    return newRefList, newDistMat


def prune_query_distance_matrix(refList, queryList, remove_seqs, qrDistMat, queryAssign=None):
    """PopPUNK/qc.py:93-135 (a contiguous block copy per kept query; done on the host: the blocks
    are already in host memory and the copy is a plain memcpy per query)."""
    remove_seqs = set(remove_seqs)
    if remove_seqs.intersection(refList):
        raise RuntimeError("Trying to remove references")
    n_ref = len(refList)
    keep = [q for q, name in enumerate(queryList) if name not in remove_seqs]
    passing_queries = [queryList[q] for q in keep]
    qr = np.asarray(qrDistMat)
    blocks = qr.reshape((len(queryList), n_ref) + qr.shape[1:])
    newqr = np.ascontiguousarray(blocks[keep]).reshape((len(keep) * n_ref,) + qr.shape[1:])
    if queryAssign is not None:
        qa = np.asarray(queryAssign)
        queryAssign = np.ascontiguousarray(
            qa.reshape((len(queryList), n_ref) + qa.shape[1:])[keep]).reshape(
                (len(keep) * n_ref,) + qa.shape[1:])
    return passing_queries, newqr, queryAssign
