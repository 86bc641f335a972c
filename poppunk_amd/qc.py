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
        # the reference takes any dtype and layout here (newDistMat = np.empty(..., dtype=distMat.dtype)): the
        # engine works on a contiguous float32 copy and the result goes back to the caller's dtype
        src = np.asarray(distMat)
        d = np.ascontiguousarray(src, dtype=np.float32)
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
        if newDistMat.dtype != src.dtype:
            newDistMat = newDistMat.astype(src.dtype)
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


# ---- distance QC (PopPUNK/qc.py:238-369,:419-468) -----------------------------------------------------

def qc_edge_lists(distMat, n_ref, max_pi_dist, max_a_dist, zeros=True, device_id=0):
    """The pairs qcDistMat looks at, as int64 [m, 2] arrays (i < j; non-self: (r, n_ref + q)):
    (core > max_pi_dist or accessory > max_a_dist, core == 0 or accessory == 0 -- None when `zeros` is
    False).  One upload of the host matrix and two mask + compaction passes on the device (`ppk_qc_edges`),
    where the reference builds two full-length Python lists of 0/1 and hands them to
    poppunk_refine.generateTuples (PopPUNK/qc.py:331-337,:348-354).  n_ref = 0: self."""
    d = np.asarray(distMat)
    if d.ndim != 2 or d.shape[1] != 2:
        raise TypeError("distMat must be an [n, 2] array")
    d = np.ascontiguousarray(d, dtype=np.float32)     # (the reference's numpy comparisons take any dtype / layout)
    lib = _lib.lib()
    llp = C.POINTER(C.c_longlong)
    modes = 3 if zeros else 1
    cap = 1 << 16
    out = np.empty((cap, 2), dtype=np.int64)
    n_edges, n_first = C.c_size_t(0), C.c_size_t(0)
    rc = lib.ppk_qc_edges(d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0], int(n_ref), modes,
                          float(max_pi_dist), float(max_a_dist), int(device_id), out.ctypes.data_as(llp), cap,
                          C.byref(n_edges), C.byref(n_first))
    if rc == _lib.ERR_CAPACITY:          # the finished lists wait on the device: fetch, do not recompute
        out = np.empty((n_edges.value, 2), dtype=np.int64)
        rc = lib.ppk_parked_fetch(out.ctypes.data_as(llp), None, None, n_edges.value, None)
    _lib.check(rc, "ppk_qc_edges")
    long_edges = out[:n_first.value]
    return long_edges, (out[n_first.value:n_edges.value] if zeros else None)


def prune_edges(long_edges, query_start, failed=None, min_count=1, allow_ref_ref=True):
    """The samples to drop so that no failing pair is left, preferring the sample with more failing
    pairs, and queries over references (PopPUNK/qc.py:419-468; same result, set of node ids).
    `long_edges`: list of (i, j) tuples or an int array [m, 2], i < j."""
    failed = set() if failed is None else failed
    e = np.asarray(long_edges, dtype=np.int64).reshape(-1, 2)
    if e.shape[0] == 0:
        return failed
    degree = np.bincount(e.ravel())
    deg = degree[e]                                      # [m, 2]: failing pairs of either end
    # worst pairs first; equal keys keep the order they came in (the reference's list.sort is stable too)
    order = np.argsort(-deg.max(axis=1), kind="stable")
    gone = np.zeros(degree.shape[0], dtype=bool)
    for v in failed:
        if 0 <= v < gone.shape[0]:
            gone[v] = True
    for (r, q), (dr, dq) in zip(e[order].tolist(), deg[order].tolist()):
        if gone[r] or gone[q] or (dr < min_count and dq < min_count):
            continue
        if r < query_start <= q:
            pick = q                                     # a reference against a query: the query goes
        elif q < query_start and not allow_ref_ref:
            continue                                     # two references, while querying: neither
        elif dr > dq and dr >= min_count:
            pick = r
        elif dq >= min_count:
            pick = q
        else:
            continue
        gone[pick] = True
        failed.add(pick)
    return failed


def qcDistMat(distMat, refList, queryList, ref_db, qc_dict, device_id=0):
    """PopPUNK/qc.py:295-369: samples whose distances are too long, or zero too often.  Returns
    (retained names in input order, {failed name: [reasons]}).  The two row predicates and the row ->
    (i, j) conversion run on the device from one upload (`qc_edge_lists`); the pruning of the few failing
    pairs is host work."""
    sys.stderr.write("Running QC on distances\n")
    sys.stderr.write("Using cutoff for core distances: " + str(qc_dict['max_pi_dist']) + "\n")
    sys.stderr.write("Using cutoff for accessory distances: " + str(qc_dict['max_a_dist']) + "\n")
    sys.stderr.write("Using cutoff for proportion of zero distances: " + str(qc_dict['prop_zero']) + "\n")
    is_self = refList == queryList
    names = refList if is_self else refList + queryList
    check_zeros = qc_dict["prop_zero"] < 1
    long_edges, zero_edges = qc_edge_lists(distMat, 0 if is_self else len(refList), qc_dict['max_pi_dist'],
                                           qc_dict['max_a_dist'], zeros=check_zeros, device_id=device_id)
    failed = prune_edges(long_edges, query_start=len(refList), allow_ref_ref=is_self)
    failed_samples = {names[x]: ["Failed distance QC (too high)"] for x in failed}
    if check_zeros:
        zero_count = round(qc_dict["prop_zero"] * len(names))
        failed = prune_edges(zero_edges, query_start=len(refList), failed=failed, min_count=zero_count,
                             allow_ref_ref=is_self)
        # (as in the reference, every sample failed so far is given this reason too: qc.py:360-366 walks the
        # whole set, not only what the zero pass added)
        for x in failed:
            failed_samples.setdefault(names[x], []).append("Failed distance QC (too many zeros)")
    dropped = frozenset(failed_samples)
    return [x for x in names if x not in dropped], failed_samples


def autoDistFind(distMat, qc_dict):
    """PopPUNK/qc.py:238-292: the core / accessory cut-offs above which distances are outliers -- the
    lowest of the top-quarter percentiles p_i that the percentile one step (1 %) further up exceeds by
    the factor y = 100*step*x/n + 1 against p_(i-step+1); the column maximum when there is no such jump.
    n = rows / r percentiles.  Returns (max_pi, max_a)."""
    d = np.asarray(distMat)
    n = int(len(d) / qc_dict['r'])
    step = int(n // 100)
    back = step - 1
    y = 100 * step * qc_dict['x'] / n + 1
    at = np.linspace(100 / n, 100, n)
    sys.stderr.write(f"Detecting maximum distance cutoffs using x = {qc_dict['x']}, r = {qc_dict['r']}\n")
    found = []
    for col, what in ((0, "core"), (1, "accessory")):
        pcs = np.percentile(d[:, col], at)
        i = np.arange(int(len(pcs) * 0.75), len(pcs) - 1)
        jumps = pcs[i][pcs[i - back] * y < pcs[i + 1]]
        if jumps.size:
            found.append(jumps.min())
        else:
            found.append(d[:, col].max())
            sys.stderr.write("No outlier detected in %s distance" % what)
    return found[0], found[1]
